#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel trace of the driver's command (20 timed frames) and the time line of its
# last burst.  Results under gpurun_out/$1/.
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/${1:-burst}
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
rm -rf "$OUT/trace"
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -- python3 "$ROOT/bench.py" --gpus 1 --steps 20 --warmup 5 --cpu-frames 0 --host-frames 0 --dropin-frames 0 --sift-frames 0 > "$OUT/bench.log" 2>&1 || { tail -5 "$OUT/bench.log"; exit 1; }
cd "$ROOT"
f=$(find "$OUT/trace" -name "*kernel_trace.csv" | head -1)
python3 tools/burst_timeline.py "$f" 300 > "$OUT/timeline.txt"
find "$OUT/trace" -name "*.csv" -size +4M -delete
head -3 "$OUT/timeline.txt"
