#!/bin/bash
# Runs on the GPU box: the driver's command (20 timed frames after 5) with the per-kernel table, and the kernels of
# the two RANSAC streams and the canvas half.  usage: tools/burst.sh <out dir under gpurun_out> [bench args]
set -u
OUT=gpurun_out/${1:-burst}; shift
mkdir -p "$OUT"
timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --cpu-frames 0 --host-frames 0 --dropin-frames 0 --sift-frames 0 \
    --kernel-table "$OUT/kernels.json" "$@" > "$OUT/bench.log" 2>&1 || { tail -5 "$OUT/bench.log"; exit 1; }
python3 - "$OUT" <<'P'
import json, sys
out = sys.argv[1]
d = json.loads([l for l in open(out + "/bench.log") if l.startswith("{")][-1])
print("value", round(d["value"]), "ms", round(d["ms_per_step"] * d["steps"], 3))
t = json.load(open(out + "/kernels.json"))["kernels"]
ks = sorted(([k, v] for k, v in t.items()), key=lambda kv: -kv[1]["total_ms"])
for k, v in ks[:14]:
    print("%-22s s%d %8.1f us/launch x %d" % (k, v["stream"], v["avg_us"], v["launches"]))
P
