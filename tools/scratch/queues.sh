#!/bin/bash
# which hardware queue each kernel of the SIFT pipeline runs on (rocprofv3 kernel trace)
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/queues; mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp; rm -rf "$OUT/trace"
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -- python3 "$ROOT/bench.py" --config 720p_sift700 --steps 96 --warmup 32 --cpu-frames 0 > "$OUT/bench.log" 2>&1 || { tail -5 "$OUT/bench.log"; exit 1; }
cd "$ROOT"
f=$(find "$OUT/trace" -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'P'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
print(rows[0].keys())
q = collections.defaultdict(collections.Counter)
for r in rows[-3000:]:
    q[r["Queue_Id"]][r["Kernel_Name"].split("(")[0][-40:]] += 1
for k, v in q.items():
    print(k, v.most_common(6))
P
find "$OUT/trace" -name "*.csv" -size +4M -delete
