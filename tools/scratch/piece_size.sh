#!/bin/bash
for cfg in "720p_orb1000" "1080p_orb1000" "720p_sift700" "360p_orb700" "4k_orb4000_canvas16k --steps 64 --warmup 8"; do
for cb in 16 24 32 16 32; do
  timeout -k 10 200 python bench.py --config $cfg --canvas-batch $cb --cpu-frames 0 --dropin-frames 0 --host-frames 0 --sift-frames 0 > gpurun_out/f.log 2>gpurun_out/f.err || { tail -3 gpurun_out/f.err; exit 1; }
  tail -1 gpurun_out/f.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print('$cfg'.split()[0], 'canvas batch $cb:', round(d['value']), 'pieces', c['canvas_pieces'], 'retries', c['canvas_half_retries'], 'void', c['frames_void'])"
done
done
