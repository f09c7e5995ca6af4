#!/usr/bin/env python3
"""Runs the bench video through Context.process_video and prints the time line of the canvas-half pieces."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rtvm = importlib.import_module("real-time-video-mosaic_amd")
import torch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 32
fh, fw, ch, cw, nf = 720, 1280, 1440, 1536, 1000
Gs, wh, ww = rtvm.synth.bounded_path(n + 1, fh, fw, seed=2)
world = rtvm.synth.make_world(wh, ww, seed=0)
dev = torch.device("cuda", 0)
frames = rtvm.synth.render_video_torch(world, Gs, fh, fw, dev)
torch.cuda.synchronize()
ctx = rtvm.Context(fh, fw, ch, cw, nfeatures=nf)
stride = fh * fw * 3
base = frames.data_ptr()
ctx.stream_begin(base, ch - fh, int(cw / 2 - fw / 2))
tr = []
t0 = time.perf_counter()
ctx.process_video([base + (1 + j) * stride for j in range(n)], batch=batch, trace=tr)
prev = t0
for t, first, cnt, done, reason in tr:
    print(f"{(t - t0) * 1e3:9.3f} ms  +{(t - prev) * 1e6:8.0f} us  frames {first:5d}..{first + cnt - 1:5d}  n={cnt:2d} done={done:2d} reason={reason}")
    prev = t
print(ctx.video_stats)
ctx.close()
