#!/bin/bash
# the column pass inside the commit (RTVM_FUSED_COMMIT=1) against the two kernels: tests, then the configurations
export RTVM_FUSED_COMMIT=1
timeout -k 10 600 python -u -m pytest tests/test_warp_blend.py -x -q -m gpu --timeout 300 --timeout-method thread > gpurun_out/fused_t.log 2>&1 || { tail -30 gpurun_out/fused_t.log; exit 1; }
tail -2 gpurun_out/fused_t.log
for v in 0 1 0 1; do
  export RTVM_FUSED_COMMIT=$v
  for cfg in "720p_orb1000" "1080p_orb1000" "4k_orb4000_canvas16k --steps 64 --warmup 8"; do
    timeout -k 10 200 python bench.py --config $cfg --cpu-frames 0 --dropin-frames 0 --host-frames 0 --sift-frames 0 > gpurun_out/f.log 2>/dev/null || exit 1
    tail -1 gpurun_out/f.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('fused $v', '$cfg'.split()[0], round(d['value']), {k: round(v,1) for k,v in d['roofline']['stream_us_per_frame'].items()})"
  done
done
