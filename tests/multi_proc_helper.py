"""One of several stitching processes that share a GPU (tests/test_multi_proc.py): its own seeded video through
Context.process_video, then a line with the statuses, the final homography and a checksum of the canvas.
usage: python multi_proc_helper.py <seed> <frames> [<frame height> <frame width> <features>]"""
import hashlib
import json
import os
import sys

import importlib

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np   # noqa: E402
import torch         # noqa: E402

rtvm = importlib.import_module("real-time-video-mosaic_amd")   # the package directory name has hyphens
synth = rtvm.synth


def run(seed, n, fh=360, fw=640, nf=500):
    ch, cw = 2 * fh, int(1.2 * fw)
    Gs, wh, ww = synth.bounded_path(n + 1, fh, fw, seed=seed)
    world = synth.make_world(wh, ww, seed=seed + 100)
    frames = synth.render_video_torch(world, Gs, fh, fw, torch.device("cuda", 0))
    torch.cuda.synchronize()
    ctx = rtvm.Context(fh, fw, ch, cw, nfeatures=nf)
    stride, base = fh * fw * 3, frames.data_ptr()
    ctx.stream_begin(base, ch - fh, int(cw / 2 - fw / 2))
    status = ctx.process_video([base + (1 + j) * stride for j in range(n)], batch=32)
    _, _, nm, H = ctx.frame_status()
    canvas = ctx.canvas_read()
    ctx.close()
    return {"seed": seed, "status": [int(v) for v in status], "matches": int(nm), "H": np.asarray(H).ravel().tolist(),
            "canvas_sha1": hashlib.sha1(np.ascontiguousarray(canvas).tobytes()).hexdigest()}


if __name__ == "__main__":
    print(json.dumps(run(*[int(v) for v in sys.argv[1:6]])))
