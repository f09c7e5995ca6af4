#!/usr/bin/env python3
"""Per-kernel HIP-event times of one rtvm_sift call and wall-clock times of the stages of a SIFT frame
(detect, knn-2 match, RANSAC, warp + blend) on synthetic 720p frames."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rtvm = importlib.import_module("real-time-video-mosaic_amd")
import torch
fh, fw, nf = (720, 1280, 700) if len(sys.argv) < 2 else (1080, 1920, 2000)
Gs, wh, ww = rtvm.synth.bounded_path(8, fh, fw, seed=2)
world = rtvm.synth.make_world(wh, ww, seed=0)
frames = list(rtvm.synth.render_video_torch(world, Gs, fh, fw, torch.device("cuda", 0)).cpu().numpy())
ctx = rtvm.Context()
ctx.sift(frames[0], nf)
t0 = time.perf_counter()
for f in frames[1:5]:
    a = ctx.sift(f, nf)
print("rtvm_sift wall %.3f ms" % ((time.perf_counter() - t0) / 4 * 1e3))
ctx.profile_start("")
b = ctx.sift(frames[5], nf)
rep = ctx.profile_stop()
tot = 0
for name, v in sorted(rep.items(), key=lambda kv: -kv[1][1]):
    tot += v[1]
    print("  %-24s %3d launches %8.3f ms" % (name, v[0], v[1]))
print("  kernels total %.3f ms" % tot)
ctx.close()
vm = rtvm.VideMosaic(frames[0], visualize=False, show_intermediate=False, nfeatures=nf)
import cProfile, pstats
pr = cProfile.Profile()
vm.process_frame(frames[1], 1)
pr.enable()
for t, f in enumerate(frames[2:], 2):
    vm.process_frame(f, t)
pr.disable()
t0 = time.perf_counter()
for t, f in enumerate(frames[2:], 8):
    vm.process_frame(f, t)
print("drop-in SIFT frame %.3f ms" % ((time.perf_counter() - t0) / len(frames[2:]) * 1e3))
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
vm.close()
