#!/usr/bin/env python3
"""Which frames of the bench video clear a canvas pixel (two non-zero pixels blending to zero: the event that
breaks the canvas masks predicted for the frames behind it in a piece)?  Stitches the video with pieces of one
frame, where nothing is ever predicted, and prints the frames.  usage: scan_cleared.py [frames] [config]"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rtvm = importlib.import_module("real-time-video-mosaic_amd")
import torch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 544
fh, fw, ch, cw, nf = 720, 1280, 1440, 1536, 1000
Gs, wh, ww = rtvm.synth.bounded_path(n + 1, fh, fw, seed=2)
world = rtvm.synth.make_world(wh, ww, seed=0)
frames = rtvm.synth.render_video_torch(world, Gs, fh, fw, torch.device("cuda", 0))
torch.cuda.synchronize()
ctx = rtvm.Context(fh, fw, ch, cw, nfeatures=nf)
stride, base = fh * fw * 3, frames.data_ptr()
ctx.stream_begin(base, ch - fh, int(cw / 2 - fw / 2))
ctx.reserve_batch(32)
cleared = []
for b0 in range(0, n, 32):
    k = min(32, n - b0)
    t = ctx.batch_detect([base + (1 + b0 + j) * stride for j in range(k)])
    for j in range(k):
        p = ctx.batch_retry(t, j, 1)
        done, reason, st, _, _ = ctx.batch_status_of(p, 1)
        assert done == 1 and st[0] == 0, (b0 + j, done, reason, st)
        if ctx.batch_cleared(p) == 0:
            cleared.append(1 + b0 + j)
    ctx.batch_release(t)
print("frames that cleared a pixel (%d of %d):" % (len(cleared), n), cleared)
ctx.close()
