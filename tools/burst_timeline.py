"""Time line of the last burst of kernels in a rocprofv3 kernel trace (…_kernel_trace.csv) of `bench.py --steps K`:
start, duration and queue of every kernel of the timed pass, relative to the first one.
usage: python tools/burst_timeline.py TRACE.csv [gap_us]"""
import csv, re, sys
rows = []
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        n = r["Kernel_Name"]
        if "k_" not in n and "rocclr" not in n:
            continue
        m = re.search(r"k_[A-Za-z0-9_]+(<[^>]*>)?|__amd_rocclr_[A-Za-z]+", n)
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), m.group(0) if m else n[:40]))
rows.sort()
gap = float(sys.argv[2]) * 1e3 if len(sys.argv) > 2 else 2e6
# the last burst: walk back from the end until a gap of more than `gap` ns
end = len(rows)
i = end - 1
while i > 0 and rows[i][0] - max(r[1] for r in rows[max(0, i - 40):i]) < gap:
    i -= 1
burst = rows[i:end]
# the timed pass of the driver's command inside it: from the bind of the last batch that was detected (k_fast)
# to the end of its canvas half
fast = [j for j, r in enumerate(rows) if r[3] == "k_fast"]
if fast:
    a = fast[-1]
    while a > 0 and rows[a][3] != "k_batch_bind":
        a -= 1
    b = a
    while b < len(rows) - 1 and not rows[b][3].startswith("k_batch_finish"):
        b += 1
    burst = rows[a:b + 1]
t0 = burst[0][0]
print("burst of %d kernels, %.1f us" % (len(burst), (max(r[1] for r in burst) - t0) / 1e3))
for s, e, q, n in burst:
    print("%9.1f  %8.1f  q%d  %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, n))
