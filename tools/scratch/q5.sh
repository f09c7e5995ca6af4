#!/bin/bash
for v in probe plain probe plain; do
  if [ $v = plain ]; then export RTVM_NO_QUEUE_PROBE=1; else unset RTVM_NO_QUEUE_PROBE; fi
  RTVM_QUEUE_DEBUG=1 timeout -k 10 300 python bench.py --procs 4 --cpu-frames 0 --host-frames 0 --dropin-frames 0 --sift-frames 0 > gpurun_out/x.log 2>gpurun_out/x.err || { tail -5 gpurun_out/x.err; exit 1; }
  grep "^rtvm:" gpurun_out/x.err | sort | uniq -c
  tail -1 gpurun_out/x.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v procs4', round(d['value']), [round(x) for x in d.get('per_process_frames_per_s', [])])"
done
