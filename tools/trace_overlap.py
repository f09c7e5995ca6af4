"""From a rocprofv3 kernel trace (…_kernel_trace.csv) of a bench run: for the last SECONDS of the trace, how long
each queue had a kernel running, how much of the main queue's busy time other queues were busy too, and the gaps
of the main queue.  usage: python tools/trace_overlap.py TRACE.csv [main-kernel-name-substring]"""
import csv
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        n = r["Kernel_Name"]
        if "k_" not in n:
            continue
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), n.split("::")[-1].split("(")[0]))
rows.sort()
# the timed pass = the last contiguous stretch; take the last 60 % of the rtvm kernels
rows = rows[int(len(rows) * 0.4):]
t0, t1 = rows[0][0], max(r[1] for r in rows)
key = sys.argv[2] if len(sys.argv) > 2 else "k_commit<false>"
mainq = next(r[2] for r in rows if key in r[3])
byq = defaultdict(list)
for s, e, q, n in rows:
    byq[q].append((s, e, n))
span = (t1 - t0) / 1e3
print("span %.1f us, main queue %d" % (span, mainq))
for q, L in sorted(byq.items()):
    busy = sum(e - s for s, e, _ in L) / 1e3
    print("queue %d: %5d kernels, busy %.1f us (%.0f %%), e.g. %s" % (q, len(L), busy, 100 * busy / span, L[0][2]))
# gaps on the main queue
L = byq[mainq]
gaps = [(L[i + 1][0] - L[i][1]) / 1e3 for i in range(len(L) - 1)]
print("main queue gaps: total %.1f us (%.0f %% of span), mean %.2f us, >20us: %d" %
      (sum(gaps), 100 * sum(gaps) / span, sum(gaps) / len(gaps), sum(g > 20 for g in gaps)))
# concurrency while the main queue runs: sweep
ev = []
for q, Lq in byq.items():
    for s, e, _ in Lq:
        ev.append((s, 1, q == mainq))
        ev.append((e, -1, q == mainq))
ev.sort()
cur_main = cur_other = 0
last = ev[0][0]
t_main_alone = t_main_with = t_other_only = t_idle = 0
for t, d, is_main in ev:
    dt = t - last
    if cur_main and cur_other:
        t_main_with += dt
    elif cur_main:
        t_main_alone += dt
    elif cur_other:
        t_other_only += dt
    else:
        t_idle += dt
    if is_main:
        cur_main += d
    else:
        cur_other += d
    last = t
tot = t_main_alone + t_main_with + t_other_only + t_idle
print("main alone %.0f %%, main + others %.0f %%, others only %.0f %%, idle %.0f %%" %
      (100 * t_main_alone / tot, 100 * t_main_with / tot, 100 * t_other_only / tot, 100 * t_idle / tot))
if len(sys.argv) > 3:
    print("largest gaps on the main queue (us, kernel before -> kernel after):")
    G = sorted(((L[i + 1][0] - L[i][1]) / 1e3, i) for i in range(len(L) - 1))[-int(sys.argv[3]):]
    for g, i in G:
        print("  %8.1f  %s -> %s   at %.1f" % (g, L[i][2], L[i + 1][2], (L[i][1] - t0) / 1e3))
