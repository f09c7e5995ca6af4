"""Host frames through the ingest ring: time against the number of frames (what is start-up, what is the rate)."""
import importlib, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
rtvm = importlib.import_module("real-time-video-mosaic_amd")
synth = rtvm.synth
fh, fw, ch, cw, nf = 720, 1280, 1440, 1536, 1000
N = 2048
Gs, wh, ww = synth.bounded_path(N + 1, fh, fw, seed=2)
world = synth.make_world(wh, ww, seed=0)
dev = torch.device("cuda", 0)
host = np.empty((N + 1, fh, fw, 3), np.uint8)
for a in range(0, N + 1, 256):
    host[a:a + 256] = synth.render_video_torch(world, Gs[a:a + 256], fh, fw, dev).cpu().numpy()
import os
dummies = [rtvm.Context(fh, fw, ch, cw, nfeatures=nf) for _ in range(int(os.environ.get("DUMMIES", "0")))]
for d in dummies:
    d.stream_begin(torch.from_numpy(host[0]).to(dev).data_ptr(), ch - fh, int(cw / 2 - fw / 2))
ctx = rtvm.Context(fh, fw, ch, cw, nfeatures=nf)
ctx.process_host_video(host[0], host[1:65], ch - fh, int(cw / 2 - fw / 2), batch=32)
for n in (512, 2048):
    t0 = time.perf_counter()
    st = ctx.process_host_video(host[0], host[1:1 + n], ch - fh, int(cw / 2 - fw / 2), batch=32)
    t = time.perf_counter() - t0
    print("n %5d  %7.2f ms  %8.1f frames/s  not ok %d  %s" % (n, t * 1e3, n / t, int((st != 0).sum()), dict(ctx.video_stats)), flush=True)
t0 = time.perf_counter(); ctx.stream_begin(ctx._ring.upload_first(host[0]), ch - fh, int(cw / 2 - fw / 2)); torch.cuda.synchronize(); print("stream_begin %.2f ms" % ((time.perf_counter() - t0) * 1e3))
ctx.close()
