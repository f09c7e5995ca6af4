#!/usr/bin/env python3
"""Workload for the rocprofv3 --pmc passes: the first N frames of the bench video (720p ORB-1000 unless a config is named) through
Context.process_video in batches of 32 with canvas pieces of 16; N = 48 lies before the first mispredicted
mask, so every kernel of the canvas half serves exactly N frames in N / 16 launches and every detection
kernel N frames in N / 32 (rounded up) launches.  usage: pmc_run.py [N] [config]"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rtvm = importlib.import_module("real-time-video-mosaic_amd")
import torch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 48
cfgs = {"720p_orb1000": (720, 1280, 1440, 1536, 1000), "4k_orb4000_canvas16k": (2160, 3840, 4096, 16384, 4000),
        "720p_sift700": (720, 1280, 1440, 1536, 700)}
name = sys.argv[2] if len(sys.argv) > 2 else "720p_orb1000"
fh, fw, ch, cw, nf = cfgs[name]
Gs, wh, ww = rtvm.synth.bounded_path(n + 1, fh, fw, seed=2)
world = rtvm.synth.make_world(wh, ww, seed=0)
frames = rtvm.synth.render_video_torch(world, Gs, fh, fw, torch.device("cuda", 0))
torch.cuda.synchronize()
det = rtvm._ffi.DET_SIFT if "sift" in name else rtvm._ffi.DET_ORB
ctx = rtvm.Context(fh, fw, ch, cw, detector=det, nfeatures=nf)
stride, base = fh * fw * 3, frames.data_ptr()
ctx.stream_begin(base, ch - fh, int(cw / 2 - fw / 2))
batch = 8 if name.startswith("4k") else 32
st = ctx.process_video([base + (1 + j) * stride for j in range(n)], batch=batch)
ctx.sync()
s = ctx.video_stats
assert not st.any() and s["retries"] == 0 and s["frames_void"] == 0, s
print("frames %d canvas_pieces %d batches %d" % (n, s["canvas_pieces"], s["batches_queued"]))
ctx.close()
