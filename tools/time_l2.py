#!/usr/bin/env python3
"""Kernel times of rtvm_match_l2_knn2 at the SIFT-2000 size (HIP events per kernel)."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rtvm = importlib.import_module("real-time-video-mosaic_amd")
import numpy as np
nq = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
nt = int(sys.argv[2]) if len(sys.argv) > 2 else nq
rng = np.random.default_rng(3)
t = rng.integers(0, 120, (nt, 128)).astype(np.float32)
q = rng.integers(0, 120, (nq, 128)).astype(np.float32)
ctx = rtvm.Context()
for _ in range(3):
    ctx.match_l2_knn2(q, t)
reps = 20
ctx.profile_start(None)
t0 = time.perf_counter()
for _ in range(reps):
    ctx.match_l2_knn2(q, t)
wall = (time.perf_counter() - t0) / reps
tab = ctx.profile_stop()
tot = 0.0
for k, (cnt, ms, _) in sorted(tab.items(), key=lambda kv: -kv[1][1]):
    print("%-28s %3d launches  %8.2f us each" % (k, cnt, ms / cnt * 1e3))
    tot += ms / reps
flop = 2.0 * nq * nt * 128
tile = tab.get("k_l2_tile<8>")
print("all kernels %.1f us per call; call incl. PCIe %.1f us" % (tot * 1e3, wall * 1e6))
if tile:
    us = tile[1] / tile[0] * 1e3
    print("k_l2_tile: %.2f us -> %.1f TFLOP/s = %.4f of 2500 (bf16 dense peak)" % (us, flop / us / 1e6, flop / us / 1e6 / 2500))
