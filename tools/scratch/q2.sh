#!/bin/bash
for d in 0 1 2 3 4 5; do
  RTVM_DUMMY_STREAMS=$d timeout -k 10 200 python bench.py --cpu-frames 0 --sift-frames 0 > gpurun_out/d$d.log 2>/dev/null || exit 1
  tail -1 gpurun_out/d$d.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('dummies $d', round(d['value']), round(d['dropin_default_flags']['orb']), round(d['host_frames_pcie_inclusive']['value']))"
done
