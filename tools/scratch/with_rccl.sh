#!/bin/bash
for v in probe plain; do
  if [ $v = plain ]; then export RTVM_NO_QUEUE_PROBE=1; else unset RTVM_NO_QUEUE_PROBE; fi
  RTVM_QUEUE_DEBUG=1 timeout -k 10 200 python tools/scratch/with_rccl.py --cpu-frames 0 --sift-frames 0 > gpurun_out/r_$v.log 2>gpurun_out/r_$v.err || { tail -5 gpurun_out/r_$v.err; exit 1; }
  grep "^rtvm:" gpurun_out/r_$v.err | sort | uniq -c
  tail -1 gpurun_out/r_$v.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v after RCCL init', round(d['value']), round(d['dropin_default_flags']['orb']), round(d['host_frames_pcie_inclusive']['value']))"
done
