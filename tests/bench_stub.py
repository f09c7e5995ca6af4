"""Stand-in for rtvm.Context when bench.py runs with RTVM_BENCH_STUB=1 (tests/test_bench_dist.py): no GPU, nothing
is stitched.  It exists so that the multi-rank plumbing of bench.py -- process group, barriers, max over ranks,
per-rank rates, the one JSON line of rank 0 -- can run end to end under torch.distributed.run on CPU.  Test
infrastructure only; the product never imports it."""
import os
import time

import numpy as np


class StubContext:
    def __init__(self, fh=0, fw=0, ch=0, cw=0, detector=0, nfeatures=700, flags=0, device=0):
        self.shape = (fh, fw, ch, cw)
        self.nfeatures = nfeatures
        self.video_stats = {}
        self._n = 0
        self._prof = False

    def reserve_batch(self, n):
        pass

    def stream_begin(self, ptr, row_off, col_off):
        self._n = 0

    def sync(self):
        pass

    def close(self):
        pass

    def process_video(self, ptrs, batch=16, trace=None, canvas_batch=None):
        n = len(ptrs)
        # rank r "stitches" at 1000 / (1 + r) frames/s: the slowest rank sets the job's time
        time.sleep(n * 1e-3 * (1 + int(os.environ.get("RANK", "0"))))
        self._n += n
        pieces = (n + batch // 2 - 1) // max(1, batch // 2)
        self.video_stats = {"frames": n, "batches_queued": (n + batch - 1) // batch, "frames_queued": n,
                            "batches_with_pending_frames": 0, "retries": 0, "frames_retried": 0, "frames_void": 0,
                            "frames_detected_again": 0, "canvas_pieces": pieces, "canvas_frames": n}
        return np.zeros(n, np.int32)

    def profile_start(self, name_filter=None):
        self._prof = True

    def profile_stop(self):
        return {"k_warp<false>": (4, 0.4, 0), "k_fast": (2, 0.1, 1), "k_rs_hyp": (2, 0.05, 2), "k_rs_refine": (2, 0.3, 3)}

    def frame_status(self):
        return 0, self.nfeatures, self.nfeatures // 2, np.eye(3)


def render_video_cpu(world, Gs, fh, fw, device):
    import torch
    # addresses only: the stand-in context never reads the frames
    return torch.zeros((len(Gs), 8), dtype=torch.uint8)


def install(rtvm):
    rtvm.Context = StubContext
    rtvm.synth.render_video_torch = render_video_cpu
    rtvm.synth.make_world = lambda h, w, seed=0, **kw: np.zeros((8, 8, 3), np.uint8)
