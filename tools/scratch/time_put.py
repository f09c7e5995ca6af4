import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
rtvm = importlib.import_module("real-time-video-mosaic_amd")
import numpy as np, torch, ctypes as C
F = rtvm._ffi
n = 1056
fh, fw, ch, cw, nf = 720, 1280, 1440, 1536, 1000
Gs, wh, ww = rtvm.synth.bounded_path(n + 1, fh, fw, seed=2)
world = rtvm.synth.make_world(wh, ww, seed=0)
host = rtvm.synth.render_video_torch(world, Gs, fh, fw, torch.device("cuda", 0)).cpu().numpy()
vm = rtvm.VideMosaic(host[0], detector_type="orb", nfeatures=nf, canvas_size=(ch, cw))
for t in range(1, 33):
    vm.process_frame(host[t], t)
vm.flush()
acc = {"glue": 0.0, "call": 0.0, "n": 0}
pipe = vm._pipe
orig_glue = vm._glue
def timed_glue(*a):
    t0 = time.perf_counter(); r = orig_glue(*a); acc["glue"] += time.perf_counter() - t0; return r
vm._glue = timed_glue
vm._pumping = True
lib = F.lib()
h = vm._ctx._h
t0 = time.perf_counter()
for t in range(33, n + 1):
    f = host[t]
    pipe.frames[pipe.n_in] = f; pipe.n_in += 1
    ta = time.perf_counter()
    rc = lib.rtvm_pipe_put(h, f.ctypes.data)
    acc["call"] += time.perf_counter() - ta
    assert rc == 0, (rc, pipe.error, lib.rtvm_last_error(h))
t1 = time.perf_counter()
vm._pumping = False
vm.flush(); vm._ctx.sync()
t2 = time.perf_counter()
k = n - 32
print("%d frames: %.0f frames/s; per frame: loop %.1f us, foreign call %.1f us of which glue %.1f us" % (k, k / (t2 - t0), (t1 - t0) / k * 1e6, acc["call"] / k * 1e6, acc["glue"] / k * 1e6))
