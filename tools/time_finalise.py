#!/usr/bin/env python3
"""Times the mosaic finalisation (crop rectangle + scale to 1920x1080) of a 1440x1536 canvas: device canvas through
librtvm.so against the CPU oracle on the same pixels."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
rtvm = importlib.import_module("real-time-video-mosaic_amd")
import oracle_ffi as orc
fh, fw = 720, 1280
first = np.random.default_rng(0).integers(30, 256, (fh, fw, 3)).astype(np.uint8)
vm = rtvm.VideMosaic(first, detector_type="orb", visualize=False, show_intermediate=False)
ctx = vm._ctx
for rep in range(3):
    t0 = time.perf_counter()
    x, y, w, h = ctx.canvas_crop_rect(15)
    x, y, w, h = x + 5, y + 5, w - 10, h - 10
    s = min(1920 / w, 1080 / h)
    out = ctx.canvas_read_scaled(x, y, w, h, int(w * s), int(h * s))
    t1 = time.perf_counter()
canvas = vm.output_img_u8()
t2 = time.perf_counter()
r = orc.crop_rect(canvas, 15)
ref = orc.resize_linear(canvas[y:y + h, x:x + w], int(w * s), int(h * s))
t3 = time.perf_counter()
print("gpu (device canvas -> host scaled image) %.3f ms; oracle (host canvas) %.1f ms; equal %s; out %s" %
      ((t1 - t0) * 1e3, (t3 - t2) * 1e3, bool(np.array_equal(out, ref)), out.shape))
vm.close()
