#!/usr/bin/env python3
"""Copies what tools/run_profiles.sh and tools/run_pmc.sh left under gpurun_out/<round>/ into profiles/ (names <round>_*)
and prints the figures the design document quotes.  usage: ROUND=r3 collect_profiles.py [config ...]"""
import glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = os.environ.get("ROUND", "r3")
SRC, DST = os.path.join(ROOT, "gpurun_out", R), os.path.join(ROOT, "profiles")


def last_json(path):
    if not os.path.exists(path):
        return None
    for line in reversed(open(path).read().splitlines()):
        if line.startswith("{"):
            return json.loads(line)
    return None


cfgs = sys.argv[1:] or ["720p_orb1000", "720p_orb1000_driver_command", "720p_sift700", "1080p_sift2000_l2", "4k_orb4000_canvas16k",
                        "1080p_orb1000", "360p_orb700"]
for c in cfgs:
    d = last_json(os.path.join(SRC, "bench_%s.log" % c))
    if d:
        json.dump(d, open(os.path.join(DST, "%s_bench_%s.json" % (R, c)), "w"), indent=1)
        r = d["roofline"]
        print("== %s: %.0f frames/s; %s %.1f us x %d launches, frac %s, traffic %s, algorithmic %s" %
              (c, d["value"], r["kernel"], r["avg_us"], r["launches"], r.get("frac"), r.get("traffic"), r.get("algorithmic_bytes")))
        for k in ("warp_blend_stage", "cpu_baseline", "cpu_baseline_all_cores", "parity_checked", "dropin_pcie_inclusive",
                  "sift_dropin_pcie_inclusive", "host_frames_pcie_inclusive"):
            if d.get(k) is not None:
                print("   %s %s" % (k, json.dumps(d[k])[:330]))
    t = last_json(os.path.join(SRC, "stats_%s.log" % c))
    if t:
        json.dump(t, open(os.path.join(DST, "%s_stats_%s_benchline.json" % (R, c)), "w"), indent=1)
        print("   traced run: %.0f frames/s, %s %.1f us x %d" % (t["value"], t["roofline"]["kernel"], t["roofline"]["avg_us"],
                                                                 t["roofline"]["launches"]))
    st = sorted(glob.glob(os.path.join(SRC, "stats_%s" % c, "*", "*_kernel_stats.csv")), key=os.path.getmtime)
    if st:
        shutil.copy(st[-1], os.path.join(DST, "%s_stats_%s.csv" % (R, c)))
    kt = os.path.join(SRC, "kernels_%s.json" % c)
    if os.path.exists(kt):
        shutil.copy(kt, os.path.join(DST, "%s_kernels_%s.json" % (R, c)))
    pm = os.path.join(SRC, "%s_pmc_traffic.json" % R if c == "720p_orb1000" else "%s_pmc_traffic_%s.json" % (R, c))
    if os.path.exists(pm):
        shutil.copy(pm, os.path.join(DST, os.path.basename(pm)))
for n in (2, 4):
    p = os.path.join(SRC, "bench_procs%d.log" % n)
    if os.path.exists(p) and "720p_orb1000" in cfgs:
        d = last_json(p)
        json.dump(d, open(os.path.join(DST, "%s_bench_720p_orb1000_procs%d.json" % (R, n)), "w"), indent=1)
        print("procs %d: %.0f aggregate, per process %s" % (n, d["value"], [round(v) for v in d["per_process_frames_per_s"]]))
