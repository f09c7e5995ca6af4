#!/bin/bash
set -e
timeout -k 10 900 python -u -m pytest tests -x -q -m gpu --timeout 300 --timeout-method thread > gpurun_out/gpu_all.log 2>&1 || { tail -30 gpurun_out/gpu_all.log; exit 1; }
tail -2 gpurun_out/gpu_all.log
for args in "--gpus 1 --steps 20 --warmup 5" "--config 720p_sift700 --cpu-frames 0" "--procs 2 --cpu-frames 0 --host-frames 0 --dropin-frames 0 --sift-frames 0" "--procs 4 --cpu-frames 0 --host-frames 0 --dropin-frames 0 --sift-frames 0" "--config 4k_orb4000_canvas16k --steps 64 --warmup 8 --cpu-frames 0 --host-frames 0 --dropin-frames 0 --sift-frames 0" "--config 1080p_orb1000 --cpu-frames 0" "--config 360p_orb700 --cpu-frames 0"; do
  timeout -k 10 300 python bench.py $args > gpurun_out/x.log 2>gpurun_out/x.err || { tail -5 gpurun_out/x.err; exit 1; }
  tail -1 gpurun_out/x.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$args', '->', round(d['value']), d.get('parity_checked'), d.get('dropin_default_flags', {}).get('orb'))"
done
