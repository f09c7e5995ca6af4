#!/bin/bash
for v in 0 1 0 1; do
  export RTVM_FUSED_COMMIT=$v
  timeout -k 10 200 python bench.py --config 540p_orb1000 --cpu-frames 0 --dropin-frames 0 --host-frames 0 --sift-frames 0 > gpurun_out/f.log 2>/dev/null || exit 1
  tail -1 gpurun_out/f.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('fused $v 540p', round(d['value']))"
done
