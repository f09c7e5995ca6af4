#!/usr/bin/env python3
"""Headless VideMosaic.process_frame at 720p: frames/s and where the host time goes (cProfile)."""
import cProfile, importlib, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rtvm = importlib.import_module("real-time-video-mosaic_amd")
import numpy as np
import torch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 544
prof = len(sys.argv) > 2 and sys.argv[2] == "prof"
fh, fw, ch, cw, nf = 720, 1280, 1440, 1536, 1000
Gs, wh, ww = rtvm.synth.bounded_path(n + 1, fh, fw, seed=2)
world = rtvm.synth.make_world(wh, ww, seed=0)
host = rtvm.synth.render_video_torch(world, Gs, fh, fw, torch.device("cuda", 0)).cpu().numpy()
vm = rtvm.VideMosaic(host[0], detector_type="orb", nfeatures=nf, canvas_size=(ch, cw), show_intermediate=False, visualize=False)
for t in range(1, 33):
    vm.process_frame(host[t], t)
vm.flush()
t0 = time.perf_counter()
a = np.empty_like(host[0])
for t in range(33, 133):
    np.copyto(a, host[t])
print("plain copy of a frame: %.1f us" % ((time.perf_counter() - t0) / 100 * 1e6))
pr = cProfile.Profile() if prof else None
t0 = time.perf_counter()
if pr: pr.enable()
for t in range(33, n + 1):
    vm.process_frame(host[t], t)
t1 = time.perf_counter()
vm.flush()
vm._ctx.sync()
if pr: pr.disable()
t2 = time.perf_counter()
print("%d frames: %.0f frames/s (enqueue loop %.1f ms, flush %.1f ms)" % (n - 32, (n - 32) / (t2 - t0), (t1 - t0) * 1e3, (t2 - t1) * 1e3))
print(vm._pipe.stats)
if pr:
    pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
