#!/bin/bash
# Runs on the GPU box: the SIFT pipeline bench with one detection stream and with two.  usage: tools/sift_ab.sh <out>
set -u
OUT=gpurun_out/${1:-sift_ab}; mkdir -p "$OUT"
for v in one two; do
    if [ $v = one ]; then export RTVM_ONE_DETECT_STREAM=1; else unset RTVM_ONE_DETECT_STREAM; fi
    timeout -k 10 300 python3 bench.py --config ${CONFIG:-720p_sift700} --cpu-frames ${CPU_FRAMES:-4} --kernel-table "$OUT/k_$v.json" > "$OUT/b_$v.log" 2>&1 || { tail -5 "$OUT/b_$v.log"; exit 1; }
    python3 - "$OUT/b_$v.log" $v <<'P'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(sys.argv[2], "detection stream(s):", round(d["value"]), "frames/s, parity", d["parity_checked"],
      {k: round(1e3 * x, 1) for k, x in d["roofline"]["stream_kernel_ms_per_frame"].items()})
P
done
