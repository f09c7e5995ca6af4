#!/bin/bash
# the column pass inside the commit (RTVM_FUSED_COMMIT=1) against the two kernels: tests, then both configurations
export RTVM_FUSED_COMMIT=1
timeout -k 10 600 python -u -m pytest tests/test_warp_blend.py tests/test_mosaic.py tests/test_headline.py -x -q -m gpu --timeout 300 --timeout-method thread > gpurun_out/fused_t.log 2>&1 || { tail -30 gpurun_out/fused_t.log; exit 1; }
tail -2 gpurun_out/fused_t.log
for v in 0 1 0 1; do
  export RTVM_FUSED_COMMIT=$v
  timeout -k 10 200 python bench.py --cpu-frames 0 --dropin-frames 0 --host-frames 0 --sift-frames 0 > gpurun_out/f.log 2>/dev/null || exit 1
  tail -1 gpurun_out/f.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('fused $v 720p', round(d['value']), d['roofline']['stream_kernel_ms_per_frame'], {k: round(v,1) for k,v in d['roofline']['stream_us_per_frame'].items()})"
  timeout -k 10 200 python bench.py --config 4k_orb4000_canvas16k --steps 64 --warmup 8 --cpu-frames 0 --dropin-frames 0 --host-frames 0 --sift-frames 0 > gpurun_out/f.log 2>/dev/null || exit 1
  tail -1 gpurun_out/f.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('fused $v 4k', round(d['value']), d['roofline']['stream_kernel_ms_per_frame'], {k: round(v,1) for k,v in d['roofline']['stream_us_per_frame'].items()})"
done
