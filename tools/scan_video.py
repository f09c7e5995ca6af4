#!/usr/bin/env python3
"""Runs the bench video through Context.process_video and prints every frame whose status is not 0."""
import importlib, os, sys
import numpy as np
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
ctx.reserve_batch(batch)
i = 1
while i <= n:
    k = min(batch, n + 1 - i)
    ctx.process_batch([base + (i + j) * stride for j in range(k)])
    done, st, nm, nk = ctx.batch_status(k)
    for j in range(k):
        if st[j] != 0:
            print("frame", i + j, "status", int(st[j]), "matches", int(nm[j]), "kp", int(nk[j]), "done", done)
    if done < k:
        print("  batch starting at", i, "consumed", done, "of", k)
    i += max(done, 1) if done < k else k
print("finished", n, "frames")
ctx.close()
