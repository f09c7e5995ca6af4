#!/bin/bash
# the default bench at several numbers of hardware queues, on one box
for q in 4 8 4 8 6; do
  GPU_MAX_HW_QUEUES=$q timeout -k 10 200 python bench.py --cpu-frames 0 > gpurun_out/q$q.log 2>/dev/null || exit 1
  tail -1 gpurun_out/q$q.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('queues $q', round(d['value']), d['dropin_default_flags']['orb'], d['dropin_default_flags']['sift'], d['host_frames_pcie_inclusive']['value'])"
done
for q in 4 8; do
  GPU_MAX_HW_QUEUES=$q timeout -k 10 200 python bench.py --steps 20 --warmup 5 --cpu-frames 0 --dropin-frames 0 --host-frames 0 --sift-frames 0 > gpurun_out/qb$q.log 2>/dev/null || exit 1
  tail -1 gpurun_out/qb$q.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('burst queues $q', round(d['value']))"
  GPU_MAX_HW_QUEUES=$q timeout -k 10 200 python bench.py --config 720p_sift700 --cpu-frames 0 > gpurun_out/qs$q.log 2>/dev/null || exit 1
  tail -1 gpurun_out/qs$q.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('sift queues $q', round(d['value']))"
done
