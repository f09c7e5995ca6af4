"""Create and destroy many stream contexts: the queue probe creates and destroys streams every time."""
import importlib, sys, time
import numpy as np
sys.path.insert(0, ".")
rtvm = importlib.import_module("real-time-video-mosaic_amd")
synth = rtvm.synth
frames, _ = synth.make_video(120, 160, 6, world_size=(400, 400), seed=1)
t0 = time.perf_counter()
last = None
for i in range(300):
    ctx = rtvm.Context(120, 160, 240, 192, nfeatures=300, detector=rtvm._ffi.DET_SIFT if i % 3 == 2 else rtvm._ffi.DET_ORB)
    st = ctx.process_host_video(frames[0], frames[1:], 120, 16, batch=4)
    c = ctx.canvas_read()
    ctx.close()
    if i % 3 == 0:
        assert last is None or np.array_equal(c, last)
        last = c
    if i % 50 == 49:
        print(i + 1, "contexts, %.1f ms each" % ((time.perf_counter() - t0) / (i + 1) * 1e3), flush=True)
print("ok")
