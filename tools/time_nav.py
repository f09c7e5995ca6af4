#!/usr/bin/env python3
"""Times the obstacle masks of analyze_for_navigation (+ the occupancy grid at scale 4) on a
1080 x 1920 picture: librtvm.so (host image in, five masks and the grid out) against the CPU
oracle on the same pixels, and the kernels alone by HIP events."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
rtvm = importlib.import_module("real-time-video-mosaic_amd")
import oracle_ffi as orc
from test_nav import scene
h, w = 1080, 1920
img = scene(h, w, 1)
boxes = [(100, 100, 300, 260), (900, 500, 1000, 640)]
ctx = rtvm.Context()
for rep in range(3):
    t0 = time.perf_counter()
    got = ctx.nav_obstacles(img, boxes, grid_scale=4)
    t1 = time.perf_counter()
ctx.profile_start("")
ctx.nav_obstacles(img, boxes, grid_scale=4)
rep = ctx.profile_stop()
t2 = time.perf_counter()
want = orc.nav_obstacles(img, boxes)
grid = orc.occupancy_grid(want["obstacles_nav"], 4)
t3 = time.perf_counter()
ok = all(np.array_equal(got[k], want[k]) for k in want) and np.array_equal(got["grid"], grid)
print("gpu call (PCIe both ways) %.3f ms; oracle %.1f ms; equal %s" % ((t1 - t0) * 1e3, (t3 - t2) * 1e3, ok))
tot = 0.0
for name, v in sorted(rep.items(), key=lambda kv: -kv[1][1]):
    tot += v[1]
    print("  %-22s %3d launches %8.3f ms" % (name, v[0], v[1]))
print("  kernels total %.3f ms" % tot)
ctx.close()
