#!/bin/bash
for v in 0 1 0 1 0 1; do
  export RTVM_FUSED_COMMIT=$v
  timeout -k 10 200 python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-frames 0 --dropin-frames 0 --host-frames 0 --sift-frames 0 > gpurun_out/f.log 2>/dev/null || exit 1
  tail -1 gpurun_out/f.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('fused $v burst', round(d['value']), d['ms_per_step']*20)"
done
for v in 0 1 0 1; do
  export RTVM_FUSED_COMMIT=$v
  timeout -k 10 200 python bench.py --config 360p_orb700 --cpu-frames 0 --dropin-frames 0 --host-frames 0 --sift-frames 0 > gpurun_out/f.log 2>/dev/null || exit 1
  tail -1 gpurun_out/f.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('fused $v 360p', round(d['value']))"
  timeout -k 10 200 python bench.py --config 720p_sift700 --cpu-frames 0 > gpurun_out/f.log 2>/dev/null || exit 1
  tail -1 gpurun_out/f.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('fused $v sift', round(d['value']))"
done
