#!/usr/bin/env python3
"""Bench video through Context.process_video: prints the canvas-half pieces that left frames pending."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rtvm = importlib.import_module("real-time-video-mosaic_amd")
import torch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 600
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 32
fh, fw, ch, cw, nf = 720, 1280, 1440, 1536, 1000
Gs, wh, ww = rtvm.synth.bounded_path(n + 1, fh, fw, seed=2)
world = rtvm.synth.make_world(wh, ww, seed=0)
frames = rtvm.synth.render_video_torch(world, Gs, fh, fw, torch.device("cuda", 0))
torch.cuda.synchronize()
ctx = rtvm.Context(fh, fw, ch, cw, nfeatures=nf)
stride, base = fh * fw * 3, frames.data_ptr()
ctx.stream_begin(base, ch - fh, int(cw / 2 - fw / 2))
tr = []
st = ctx.process_video([base + (1 + j) * stride for j in range(n)], batch=batch, trace=tr)
for t, first, cnt, done, reason in tr:
    if reason:
        print("piece of frames %d..%d: consumed %d, reason %d" % (first, first + cnt - 1, done, reason))
print("statuses not 0:", [(i, int(s)) for i, s in enumerate(st) if s])
print(ctx.video_stats)
ctx.close()
