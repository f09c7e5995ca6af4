"""ctypes binding of librtvm.so (C ABI declared in include/rtvm.h).

There is no CPU fallback: if the library is missing, or no HIP device can be
opened, the calls raise.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# RTVM_LIBRARY=<path> loads another build of the library (A/B runs of two builds on one box)
LIB_PATH = os.environ.get("RTVM_LIBRARY") or os.path.join(_HERE, "lib", "librtvm.so")
# Canvas-half pieces rtvm_process_video keeps queued for frames resident on the device (one more when the
# frames come from host memory through the ingest ring): the outcome of a piece is read while the next one runs.  A
# third one only adds a piece that did nothing (and its launches) to every mispredicted canvas mask: measured
# 17 450 frames/s with three, 17 720 with two, 17 350 with four (720p ORB-1000).
PIECES_IN_FLIGHT = 2

OK = 0
E_ARG, E_HIP, E_NOMEM, E_DEGENERATE, E_STATE = -1, -2, -3, -4, -5
DET_ORB, DET_SIFT_MATCH_ONLY, DET_SIFT = 0, 1, 2
FLAG_WARP_FIXED = 1
FLAG_MATCH_LEGACY = 2
FLAG_SINGLE_STREAM = 4
FLAG_GRAY_LEGACY = 8
FRAME_FEW_MATCHES, FRAME_NO_HOMOGRAPHY, FRAME_INVALID_H = 1, 2, 4
PENDING_NONE, PENDING_SKIPPED_FRAME, PENDING_MASK, PENDING_VOID = 0, 1, 2, 3

# name -> (restype, argtypes); also the list the symbol-export test walks
_u8p = C.POINTER(C.c_uint8)
_i32p = C.POINTER(C.c_int32)
_f32p = C.POINTER(C.c_float)
_f64p = C.POINTER(C.c_double)
_intp = C.POINTER(C.c_int)
_vp = C.c_void_p
SIGNATURES = {
    "rtvm_version": (C.c_int, []),
    "rtvm_create": (C.c_int, [C.POINTER(_vp)] + [C.c_int] * 8),
    "rtvm_destroy": (None, [_vp]),
    "rtvm_last_error": (C.c_char_p, [_vp]),
    "rtvm_sync": (C.c_int, [_vp]),
    "rtvm_max_keypoints": (C.c_int, [_vp]),
    "rtvm_gray": (C.c_int, [_vp, _vp, _vp]),
    "rtvm_orb": (C.c_int, [_vp, _vp, _intp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "rtvm_match_hamming_xcheck": (C.c_int, [_vp, _vp, C.c_int, _vp, C.c_int, _vp, _vp, _vp, _intp]),
    "rtvm_match_hamming_xcheck_sorted": (C.c_int, [_vp, _vp, C.c_int, _vp, C.c_int, _vp, _vp, _vp, _intp]),
    "rtvm_match_hamming_xcheck_noalloc": (C.c_int, [_vp, C.c_int, _vp, C.c_int, _vp, _vp, _vp, _vp, C.c_int, _vp]),
    "rtvm_noalloc_last_error": (C.c_char_p, []),
    "rtvm_match_l2_knn2": (C.c_int, [_vp, _vp, C.c_int, _vp, C.c_int, C.c_int, _vp, _vp]),
    "rtvm_find_homography_ransac": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_double, C.c_int, C.c_double, _vp, _vp, _intp]),
    "rtvm_validate_homography": (C.c_int, [_vp, _vp, C.c_double, C.c_double, _intp]),
    "rtvm_canvas_init": (C.c_int, [_vp, _vp, C.c_int, C.c_int]),
    "rtvm_warp_perspective": (C.c_int, [_vp, _vp, _vp, _vp]),
    "rtvm_distance_transform": (C.c_int, [_vp, _vp, C.c_int, C.c_int, _vp]),
    "rtvm_warp_blend": (C.c_int, [_vp, _vp, _vp]),
    "rtvm_canvas_read": (C.c_int, [_vp, _vp]),
    "rtvm_canvas_read_f64": (C.c_int, [_vp, _vp]),
    "rtvm_sift_max_keypoints": (C.c_int, [C.c_int]),
    "rtvm_sift": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, _intp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "rtvm_ingest_reserve": (C.c_int, [_vp, C.c_int]),
    "rtvm_ingest_host_slot": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_void_p)]),
    "rtvm_ingest_write": (C.c_int, [_vp, C.c_int, _vp]),
    "rtvm_ingest_submit": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_void_p)]),
    "rtvm_ingest_submit_range": (C.c_int, [_vp, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "rtvm_crop_rect": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, _vp, _intp]),
    "rtvm_canvas_crop_rect": (C.c_int, [_vp, C.c_int, _vp, _intp]),
    "rtvm_resize_linear": (C.c_int, [_vp, _vp, C.c_int, C.c_int, _vp, C.c_int, C.c_int]),
    "rtvm_canvas_read_scaled": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, C.c_int, C.c_int]),
    "rtvm_bgr2hsv": (C.c_int, [_vp, _vp, C.c_int, C.c_int, _vp]),
    "rtvm_morphology": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp]),
    "rtvm_canny": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_double, C.c_double, _vp]),
    "rtvm_nav_obstacles": (C.c_int, [_vp, _vp, C.c_int, C.c_int, _vp, C.c_int, _vp, _vp, _vp, _vp, _vp, C.c_int, _vp]),
    "rtvm_occupancy_grid": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, _vp]),
    "rtvm_stream_begin": (C.c_int, [_vp, _vp, C.c_int, C.c_int]),
    "rtvm_process_frame": (C.c_int, [_vp, _vp]),
    "rtvm_frame_status": (C.c_int, [_vp, _intp, _intp, _intp, _vp]),
    "rtvm_reserve_batch": (C.c_int, [_vp, C.c_int]),
    "rtvm_process_batch": (C.c_int, [_vp, _vp, C.c_int]),
    "rtvm_batch_status": (C.c_int, [_vp, _intp, _vp, _vp, _vp, C.c_int]),
    "rtvm_batch_status_at": (C.c_int, [_vp, C.c_int, _intp, _vp, _vp, _vp, C.c_int]),
    "rtvm_batch_ticket": (C.c_long, [_vp]),
    "rtvm_batch_status_of": (C.c_int, [_vp, C.c_long, _intp, _intp, _vp, _vp, _vp, C.c_int]),
    "rtvm_batch_cleared": (C.c_int, [_vp, C.c_long, _intp]),
    "rtvm_batch_retry": (C.c_int, [_vp, C.c_long, C.c_int, C.c_int]),
    "rtvm_batch_detect": (C.c_int, [_vp, _vp, C.c_int]),
    "rtvm_batch_release": (C.c_int, [_vp, C.c_long]),
    "rtvm_batch_results": (C.c_int, [_vp, C.c_long, _vp, _vp, _vp, _vp, C.c_int]),
    "rtvm_batch_results_ready": (C.c_int, [_vp, C.c_long]),
    "rtvm_batch_commit": (C.c_int, [_vp, C.c_long, C.c_int, C.c_int, _vp]),
    "rtvm_batch_done": (C.c_int, [_vp, C.c_long]),
    "rtvm_batch_rebase": (C.c_int, [_vp]),
    "rtvm_batch_get_features": (C.c_int, [_vp, C.c_long, C.c_int, _intp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "rtvm_batch_get_matches": (C.c_int, [_vp, C.c_long, C.c_int, _intp, _vp, _vp, _vp]),
    "rtvm_process_video": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, _vp, _vp]),
    "rtvm_process_video_source": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, C.c_int, _intp]),
    "rtvm_pipe_open": (C.c_int, [_vp, C.c_int, C.c_int, _vp, _vp]),
    "rtvm_pipe_put": (C.c_int, [_vp, _vp]),
    "rtvm_pipe_flush": (C.c_int, [_vp]),
    "rtvm_pipe_busy": (C.c_int, [_vp]),
    "rtvm_pipe_last": (C.c_int, [_vp, C.POINTER(C.c_long), _intp]),
    "rtvm_pipe_get_stats": (C.c_int, [_vp, _vp]),
    "rtvm_pipe_close": (C.c_int, [_vp]),
    "rtvm_first_frame": (C.c_int, [_vp, _vp, C.c_int, C.c_int]),
    "rtvm_frame_register": (C.c_int, [_vp, _vp, _intp, _intp, _vp]),
    "rtvm_frame_commit": (C.c_int, [_vp, _vp]),
    "rtvm_stream_set_prev": (C.c_int, [_vp, _vp]),
    "rtvm_get_features": (C.c_int, [_vp, C.c_int, _intp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "rtvm_get_matches": (C.c_int, [_vp, _intp, _vp, _vp, _vp]),
    "rtvm_profile_start": (C.c_int, [_vp, C.c_char_p]),
    "rtvm_profile_stop": (C.c_int, [_vp]),
    "rtvm_profile_report": (C.c_int, [_vp, C.c_char_p, C.c_int]),
}



class VideoStats(C.Structure):
    """rtvm_video_stats"""
    _fields_ = [(k, C.c_int) for k in ("frames", "batches_queued", "frames_queued", "batches_with_pending_frames", "retries",
                                       "frames_retried", "frames_void", "frames_detected_again", "canvas_pieces",
                                       "canvas_frames")]


class VideoEvent(C.Structure):
    """rtvm_video_event"""
    _fields_ = [("seconds", C.c_double), ("first", C.c_int), ("count", C.c_int), ("done", C.c_int), ("reason", C.c_int)]


class PipeStats(C.Structure):
    """rtvm_pipe_stats"""
    _fields_ = [(k, C.c_long) for k in ("frames", "batches", "pieces", "pieces_with_pending_frames", "pieces_void",
                                        "frames_detected_again")]


FRAME_SOURCE = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p))   # rtvm_frame_source
GLUE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_void_p, C.c_void_p)   # rtvm_glue_fn (the matrices as addresses)
E_CALLBACK = -6

_lib = None
HIP_RUNTIME = None   # path of the HIP runtime preloaded for librtvm.so (None: the linker's choice)


def _preload_hip_runtime():
    """One process can host only one HIP runtime.  PyTorch's wheel ships its own
    libamdhip64.so (same SONAME as the system one), so when torch is installed its copy
    is loaded first and librtvm.so binds to it; a later `import torch` then finds the
    very same file.  RTVM_HIP_RUNTIME=<path> overrides, RTVM_HIP_RUNTIME=system skips."""
    global HIP_RUNTIME
    choice = os.environ.get("RTVM_HIP_RUNTIME", "")
    if choice == "system":
        return
    path = choice
    if not path:
        import importlib.util
        try:
            spec = importlib.util.find_spec("torch")
        except (ImportError, ValueError):
            spec = None
        if spec is not None and spec.submodule_search_locations:
            cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
            if os.path.exists(cand):
                path = cand
    if path:
        C.CDLL(path, mode=C.RTLD_GLOBAL)
        HIP_RUNTIME = path


def lib():
    """Load librtvm.so once; raise if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "librtvm.so not found at %s: build it with `make lib` "
                "(there is no CPU fallback)" % LIB_PATH)
        _preload_hip_runtime()
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


class RtvmError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("rtvm error %d: %s" % (code, msg))
        self.code = code


def ptr(a):
    """void* of a C-contiguous numpy array (or None)."""
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)


NOALLOC_MAX_DESCRIPTORS = 16384


def match_hamming_noalloc(q_dev, nq, t_dev, nt, qidx_dev, tidx_dev, dist_dev, m_out_dev, sorted_by_distance=False,
                          stream=0):
    """rtvm_match_hamming_xcheck_noalloc: device addresses in and out (integers), queued on the HIP
    stream `stream` of the current device; returns without waiting.  Allocates nothing."""
    rc = lib().rtvm_match_hamming_xcheck_noalloc(C.c_void_p(q_dev), int(nq), C.c_void_p(t_dev), int(nt),
                                                 C.c_void_p(qidx_dev), C.c_void_p(tidx_dev), C.c_void_p(dist_dev),
                                                 C.c_void_p(m_out_dev), int(bool(sorted_by_distance)), C.c_void_p(stream))
    if rc != OK:
        raise RtvmError(rc, lib().rtvm_noalloc_last_error().decode("utf-8", "replace"))


class HostFrameRing:
    """Frame source for Context.process_video: frame i of `frames` (host arrays) is copied into
    pinned slot i % n_slots and from there to the device when its batch is about to be detected.
    n_slots must cover the frames process_video keeps alive (four batches) plus the batch that is
    copied ahead; one more slot holds the first frame of the stream.  The copies into the pinned
    slots are rtvm_ingest_write calls (a few threads inside the library, non-temporal stores: about
    30 us for a 720p frame) made from one background thread, so that the batch after the one just
    handed out is copied while the device works."""

    def __init__(self, ctx, n_slots):
        from concurrent.futures import ThreadPoolExecutor
        ctx.ingest_reserve(n_slots + 1)
        self.ctx, self.n_slots, self.frames = ctx, n_slots, []
        self.pool = ThreadPoolExecutor(max_workers=1)
        self.ahead = None   # (first, k, futures) of the batch being copied ahead

    def __len__(self):
        return len(self.frames)

    def upload_first(self, frame):
        self._settle()
        self.ctx.ingest_write(self.n_slots, frame)
        return self.ctx.ingest_submit(self.n_slots)

    def _copy(self, first, k):
        frames, n, write = self.frames, self.n_slots, self.ctx.ingest_write

        def job():
            for i in range(first, first + k):
                write(i % n, frames[i])
        return [self.pool.submit(job)]

    def _settle(self):
        if self.ahead is not None:
            for f in self.ahead[2]:
                f.result()
            self.ahead = None

    def acquire(self, first, k):
        if self.ahead is not None and self.ahead[:2] == (first, k):
            futures, self.ahead = self.ahead[2], None
        else:
            self._settle()   # a different range is wanted (detection started over): let the stray copy finish
            futures = self._copy(first, k)
        for f in futures:
            f.result()
        s0 = first % self.n_slots
        if s0 + k <= self.n_slots:   # consecutive slots: one copy for the whole batch
            ptrs = self.ctx.ingest_submit_range(s0, k)
        else:
            ptrs = [self.ctx.ingest_submit(i % self.n_slots) for i in range(first, first + k)]
        # the next batch, if its slots are not those of a batch that may still be alive (four before it)
        nxt, k2 = first + k, min(k, len(self.frames) - first - k)
        if k2 > 0 and 5 * k <= self.n_slots:
            self.ahead = (nxt, k2, self._copy(nxt, k2))
        return ptrs


def _c_u8(image):
    """The image as a C-contiguous H x W x 3 u8 array (the reference converts with astype(uint8) too)."""
    image = np.ascontiguousarray(image, dtype=np.uint8)
    if image.ndim != 3 or image.shape[2] != 3:
        raise ValueError("expected an H x W x 3 image")
    return image


class Context:
    """Owns one rtvm_ctx (one video stream on one GPU)."""

    def __init__(self, frame_h=0, frame_w=0, canvas_h=0, canvas_w=0, detector=DET_ORB,
                 nfeatures=700, flags=0, device=0):
        self._h = C.c_void_p()
        L = lib()
        rc = L.rtvm_create(C.byref(self._h), device, frame_h, frame_w, canvas_h, canvas_w,
                           detector, nfeatures, flags)
        if rc != OK:
            self._h = None
            raise RtvmError(rc, "rtvm_create failed (no HIP device, bad size or out of memory)")
        self.frame_h, self.frame_w = frame_h, frame_w
        self.canvas_h, self.canvas_w = canvas_h, canvas_w
        self.nfeatures = nfeatures
        self.device = device
        self.detector = detector

    def close(self):
        if getattr(self, "_h", None):
            lib().rtvm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != OK:
            raise RtvmError(rc, lib().rtvm_last_error(self._h).decode("utf-8", "replace"))

    # ---- stages ---------------------------------------------------------
    def max_keypoints(self):
        return lib().rtvm_max_keypoints(self._h)

    def gray(self, bgr):
        bgr = self._frame(bgr, "gray")
        out = np.empty(bgr.shape[:2], np.uint8)
        self._check(lib().rtvm_gray(self._h, ptr(bgr), ptr(out)))
        return out

    def orb(self, bgr):
        bgr = self._frame(bgr, "orb")
        cap = self.max_keypoints()
        n = C.c_int(0)
        xy = np.empty((cap, 2), np.float32)
        size = np.empty(cap, np.float32)
        angle = np.empty(cap, np.float32)
        resp = np.empty(cap, np.float32)
        octave = np.empty(cap, np.int32)
        desc = np.empty((cap, 32), np.uint8)
        self._check(lib().rtvm_orb(self._h, ptr(bgr), C.byref(n), ptr(xy), ptr(size), ptr(angle),
                                   ptr(resp), ptr(octave), ptr(desc)))
        k = n.value
        return dict(xy=xy[:k].copy(), size=size[:k].copy(), angle=angle[:k].copy(),
                    response=resp[:k].copy(), octave=octave[:k].copy(), desc=desc[:k].copy())

    def match_hamming(self, q, t, sorted_by_distance=False):
        q = np.ascontiguousarray(q, dtype=np.uint8).reshape(-1, 32)
        t = np.ascontiguousarray(t, dtype=np.uint8).reshape(-1, 32)
        nq, nt = len(q), len(t)
        qi = np.empty(max(nq, 1), np.int32)
        ti = np.empty(max(nq, 1), np.int32)
        di = np.empty(max(nq, 1), np.int32)
        m = C.c_int(0)
        fn = lib().rtvm_match_hamming_xcheck_sorted if sorted_by_distance else lib().rtvm_match_hamming_xcheck
        self._check(fn(self._h, ptr(q), nq, ptr(t), nt, ptr(qi), ptr(ti), ptr(di), C.byref(m)))
        k = m.value
        return qi[:k].copy(), ti[:k].copy(), di[:k].copy()

    def match_l2_knn2(self, q, t):
        q = np.ascontiguousarray(q, dtype=np.float32)
        t = np.ascontiguousarray(t, dtype=np.float32)
        nq, dim = q.shape
        nt = t.shape[0]
        idx = np.empty((nq, 2), np.int32)
        dist = np.empty((nq, 2), np.float32)
        self._check(lib().rtvm_match_l2_knn2(self._h, ptr(q), nq, ptr(t), nt, dim, ptr(idx), ptr(dist)))
        return idx, dist

    def find_homography(self, src_xy, dst_xy, thresh=2.0, max_iters=2000, confidence=0.995):
        src = np.ascontiguousarray(src_xy, dtype=np.float32).reshape(-1, 2)
        dst = np.ascontiguousarray(dst_xy, dtype=np.float32).reshape(-1, 2)
        m = len(src)
        H = np.zeros(9, np.float64)
        mask = np.zeros(max(m, 1), np.uint8)
        found = C.c_int(0)
        self._check(lib().rtvm_find_homography_ransac(self._h, ptr(src), ptr(dst), m, thresh, max_iters,
                                                      confidence, ptr(H), ptr(mask), C.byref(found)))
        if not found.value:
            return None, mask[:m]
        return H.reshape(3, 3), mask[:m]

    def _frame(self, bgr, who):
        """The frame as the C side reads it: C-contiguous u8 of exactly frame_h x frame_w x 3."""
        bgr = np.ascontiguousarray(bgr, dtype=np.uint8)
        if bgr.shape != (self.frame_h, self.frame_w, 3):
            raise ValueError("%s: frame of shape %s, the context was created for %s"
                             % (who, bgr.shape, (self.frame_h, self.frame_w, 3)))
        return bgr

    def validate_homography(self, H, translation_threshold=50.0, scale_threshold=0.3):
        """validate_homography (main.py:761-801) as the device-resident stream path evaluates it."""
        H = np.ascontiguousarray(H, dtype=np.float64).reshape(9)
        ok = C.c_int(0)
        self._check(lib().rtvm_validate_homography(self._h, ptr(H), float(translation_threshold),
                                                   float(scale_threshold), C.byref(ok)))
        return bool(ok.value)

    def canvas_init(self, first_bgr, row_off, col_off):
        first_bgr = self._frame(first_bgr, "canvas_init")
        self._check(lib().rtvm_canvas_init(self._h, ptr(first_bgr), int(row_off), int(col_off)))

    def warp_perspective(self, bgr, H):
        bgr = self._frame(bgr, "warp_perspective")
        H = np.ascontiguousarray(H, dtype=np.float64).reshape(9)
        out = np.empty((self.canvas_h, self.canvas_w, 3), np.uint8)
        self._check(lib().rtvm_warp_perspective(self._h, ptr(bgr), ptr(H), ptr(out)))
        return out

    def distance_transform(self, mask):
        mask = np.ascontiguousarray(mask, dtype=np.uint8)
        h, w = mask.shape
        out = np.empty((h, w), np.float32)
        self._check(lib().rtvm_distance_transform(self._h, ptr(mask), h, w, ptr(out)))
        return out

    def warp_blend(self, bgr, H):
        bgr = self._frame(bgr, "warp_blend")
        H = np.ascontiguousarray(H, dtype=np.float64).reshape(9)
        self._check(lib().rtvm_warp_blend(self._h, ptr(bgr), ptr(H)))

    def canvas_read(self):
        out = np.empty((self.canvas_h, self.canvas_w, 3), np.uint8)
        self._check(lib().rtvm_canvas_read(self._h, ptr(out)))
        return out

    def canvas_read_f64(self):
        """The mosaic as float64 (the reference's dtype, main.py:80-81): widened on the host from the bytes that
        crossed PCIe."""
        out = np.empty((self.canvas_h, self.canvas_w, 3), np.float64)
        self._check(lib().rtvm_canvas_read_f64(self._h, C.c_void_p(out.ctypes.data)))
        return out

    def sync(self):
        self._check(lib().rtvm_sync(self._h))

    def sift(self, bgr, nfeatures=700):
        """cv2.SIFT_create(nfeatures).detectAndCompute on a BGR u8 frame: dict of arrays."""
        bgr = _c_u8(bgr)
        cap = lib().rtvm_sift_max_keypoints(int(nfeatures))
        n = C.c_int(0)
        xy = np.empty((cap, 2), np.float32)
        size = np.empty(cap, np.float32)
        angle = np.empty(cap, np.float32)
        resp = np.empty(cap, np.float32)
        octave = np.empty(cap, np.int32)
        desc = np.empty((cap, 128), np.float32)
        self._check(lib().rtvm_sift(self._h, ptr(bgr), bgr.shape[0], bgr.shape[1], int(nfeatures), C.byref(n), ptr(xy),
                                    ptr(size), ptr(angle), ptr(resp), ptr(octave), ptr(desc)))
        k = n.value
        return dict(xy=xy[:k].copy(), size=size[:k].copy(), angle=angle[:k].copy(), response=resp[:k].copy(),
                    octave=octave[:k].copy(), desc=desc[:k].copy())

    # ---- mosaic finalisation (main.py:980-1038) -------------------------------
    def crop_rect(self, image, threshold=15):
        """cv2.boundingRect of the pixels of a BGR u8 image whose gray value exceeds threshold:
        (x, y, w, h), or None when there is none."""
        image = _c_u8(image)
        rect = np.zeros(4, np.int32)
        found = C.c_int(0)
        self._check(lib().rtvm_crop_rect(self._h, ptr(image), image.shape[0], image.shape[1], int(threshold), ptr(rect),
                                         C.byref(found)))
        return tuple(int(v) for v in rect) if found.value else None

    def canvas_crop_rect(self, threshold=15):
        """crop_rect of the canvas as it stands on the device."""
        rect = np.zeros(4, np.int32)
        found = C.c_int(0)
        self._check(lib().rtvm_canvas_crop_rect(self._h, int(threshold), ptr(rect), C.byref(found)))
        return tuple(int(v) for v in rect) if found.value else None

    def resize_linear(self, image, new_w, new_h):
        """cv2.resize(image, (new_w, new_h), interpolation=cv2.INTER_LINEAR) of a BGR u8 image."""
        image = _c_u8(image)
        out = np.empty((int(new_h), int(new_w), 3), np.uint8)
        self._check(lib().rtvm_resize_linear(self._h, ptr(image), image.shape[0], image.shape[1], ptr(out), int(new_h),
                                             int(new_w)))
        return out

    def canvas_read_scaled(self, x, y, w, h, new_w, new_h):
        """The rectangle x, y, w, h of the device canvas, resized like resize_linear."""
        out = np.empty((int(new_h), int(new_w), 3), np.uint8)
        self._check(lib().rtvm_canvas_read_scaled(self._h, int(x), int(y), int(w), int(h), ptr(out), int(new_h),
                                                  int(new_w)))
        return out

    # ---- classical passes before path planning (main.py:1051-1232, 1344-1359) ----
    def bgr2hsv(self, image):
        """cv2.cvtColor(image, cv2.COLOR_BGR2HSV) of a BGR u8 image."""
        image = _c_u8(image)
        out = np.empty_like(image)
        self._check(lib().rtvm_bgr2hsv(self._h, ptr(image), image.shape[0], image.shape[1], ptr(out)))
        return out

    def morphology(self, mask, op, shape, ksize, iterations=1):
        """cv2.erode / dilate / morphologyEx(OPEN, CLOSE) of a 0/255 mask with
        getStructuringElement(shape, (ksize, ksize)); op 0..3 = cv2.MORPH_ERODE..MORPH_CLOSE,
        shape 0..2 = cv2.MORPH_RECT, MORPH_CROSS, MORPH_ELLIPSE."""
        mask = np.ascontiguousarray(mask, dtype=np.uint8)
        assert mask.ndim == 2
        out = np.empty_like(mask)
        self._check(lib().rtvm_morphology(self._h, ptr(mask), mask.shape[0], mask.shape[1], int(op), int(shape),
                                          int(ksize), int(iterations), ptr(out)))
        return out

    def canny(self, gray, threshold1, threshold2):
        """cv2.Canny(gray, threshold1, threshold2) of a u8 image (aperture 3, L1 gradient): edges 0 / 255."""
        gray = np.ascontiguousarray(gray, dtype=np.uint8)
        if gray.ndim != 2:
            raise ValueError("canny: a single-channel image is expected, got shape %r" % (gray.shape,))
        out = np.empty_like(gray)
        self._check(lib().rtvm_canny(self._h, ptr(gray), gray.shape[0], gray.shape[1], float(threshold1),
                                     float(threshold2), ptr(out)))
        return out

    def nav_obstacles(self, image, boxes=(), grid_scale=None):
        """The obstacle masks of analyze_for_navigation for a BGR u8 image and filled rectangles
        (x1, y1, x2, y2, both corners included): dict with mask_fire, mask_smoke, texture_mask,
        obstacles, obstacles_nav and, with grid_scale, the occupancy grid."""
        image = _c_u8(image)
        h, w = image.shape[:2]
        boxes = np.ascontiguousarray(np.asarray(boxes, dtype=np.int32).reshape(-1, 4))
        names = ("mask_fire", "mask_smoke", "texture_mask", "obstacles", "obstacles_nav")
        out = {k: np.empty((h, w), np.uint8) for k in names}
        grid = None
        if grid_scale is not None:
            grid = np.empty((max(1, h // int(grid_scale)), max(1, w // int(grid_scale))), np.uint8)
        self._check(lib().rtvm_nav_obstacles(self._h, ptr(image), h, w, ptr(boxes) if len(boxes) else None, len(boxes),
                                             *[ptr(out[k]) for k in names], int(grid_scale or 0),
                                             ptr(grid) if grid is not None else None))
        if grid is not None:
            out["grid"] = grid
        return out

    def occupancy_grid(self, mask, scale=4):
        """find_path_astar's matrix: 1 where more than 30 % of a scale x scale block is set."""
        mask = np.ascontiguousarray(mask, dtype=np.uint8)
        h, w = mask.shape
        grid = np.empty((max(1, h // int(scale)), max(1, w // int(scale))), np.uint8)
        self._check(lib().rtvm_occupancy_grid(self._h, ptr(mask), h, w, int(scale), ptr(grid)))
        return grid

    # ---- two-call stream form (host frames, glue on the host) ---------------
    def first_frame(self, bgr, row_off, col_off):
        bgr = self._frame(bgr, "first_frame")
        self._check(lib().rtvm_first_frame(self._h, ptr(bgr), int(row_off), int(col_off)))

    def frame_register(self, bgr):
        bgr = self._frame(bgr, "frame_register")
        m, found = C.c_int(0), C.c_int(0)
        H = np.zeros(9, np.float64)
        self._check(lib().rtvm_frame_register(self._h, ptr(bgr), C.byref(m), C.byref(found), ptr(H)))
        return m.value, bool(found.value), H.reshape(3, 3)

    def stream_set_prev(self, bgr):
        """Makes the features of this host frame kp_prev/des_prev of the stream (nothing else changes)."""
        bgr = self._frame(bgr, "stream_set_prev")
        self._check(lib().rtvm_stream_set_prev(self._h, ptr(bgr)))

    def frame_commit(self, H):
        H = np.ascontiguousarray(H, dtype=np.float64).reshape(9)
        self._check(lib().rtvm_frame_commit(self._h, ptr(H)))

    def _desc_buffer(self, cap):
        """Descriptor array of the stream path: 32 bytes per keypoint (ORB) or 128 floats (DET_SIFT)."""
        return np.empty((cap, 128), np.float32) if self.detector == DET_SIFT else np.empty((cap, 32), np.uint8)

    def _distances(self, di):
        """Match distances as the reference's DMatch carries them: Hamming counts, or float L2 distances (DET_SIFT:
        the library hands over their bit patterns)."""
        return di.view(np.float32) if self.detector == DET_SIFT else di

    def get_features(self, slot):
        cap = max(self.max_keypoints(), 1)
        n = C.c_int(0)
        xy = np.empty((cap, 2), np.float32)
        size = np.empty(cap, np.float32)
        angle = np.empty(cap, np.float32)
        resp = np.empty(cap, np.float32)
        octave = np.empty(cap, np.int32)
        desc = self._desc_buffer(cap)
        self._check(lib().rtvm_get_features(self._h, int(slot), C.byref(n), ptr(xy), ptr(size), ptr(angle),
                                            ptr(resp), ptr(octave), ptr(desc)))
        k = n.value
        return dict(xy=xy[:k].copy(), size=size[:k].copy(), angle=angle[:k].copy(),
                    response=resp[:k].copy(), octave=octave[:k].copy(), desc=desc[:k].copy())

    def get_matches(self):
        cap = max(self.max_keypoints(), 1)
        m = C.c_int(0)
        qi = np.empty(cap, np.int32)
        ti = np.empty(cap, np.int32)
        di = np.empty(cap, np.int32)
        self._check(lib().rtvm_get_matches(self._h, C.byref(m), ptr(qi), ptr(ti), ptr(di)))
        k = m.value
        return qi[:k].copy(), ti[:k].copy(), self._distances(di[:k].copy())

    # ---- per-kernel timing ----------------------------------------------------
    def profile_start(self, name_filter=None):
        self._check(lib().rtvm_profile_start(self._h, name_filter.encode() if name_filter else None))

    def profile_stop(self):
        """Returns {kernel name: (launches, total_ms, stream)}; stream 0 = main (canvas half),
        1 = detection + matching, 2 = RANSAC search, 3 = RANSAC polish."""
        self._check(lib().rtvm_profile_stop(self._h))
        n = lib().rtvm_profile_report(self._h, None, 0)
        buf = C.create_string_buffer(n + 16)
        lib().rtvm_profile_report(self._h, buf, n + 16)
        out = {}
        for line in buf.value.decode().splitlines():
            name, cnt, ms, stream = line.rsplit(" ", 3)
            out[name] = (int(cnt), float(ms), int(stream))
        return out

    # ---- device-resident stream path -------------------------------------
    def stream_begin(self, bgr_dev_ptr, row_off, col_off):
        self._check(lib().rtvm_stream_begin(self._h, C.c_void_p(bgr_dev_ptr), int(row_off), int(col_off)))

    def process_frame(self, bgr_dev_ptr):
        self._check(lib().rtvm_process_frame(self._h, C.c_void_p(bgr_dev_ptr)))

    def reserve_batch(self, n):
        self._check(lib().rtvm_reserve_batch(self._h, int(n)))

    def process_batch(self, bgr_dev_ptrs):
        """Queue process_frame for consecutive device-resident frames (list of device pointers)."""
        n = len(bgr_dev_ptrs)
        arr = (C.c_void_p * n)(*bgr_dev_ptrs)
        self._check(lib().rtvm_process_batch(self._h, arr, n))

    def batch_status(self, n, back=0):
        """(frames consumed, status[n], matches[n], keypoints[n]) of the most recent batch
        (back=0) or of the ones queued before it (back=1, 2).  Waits for that batch."""
        done = C.c_int(0)
        st = np.zeros(n, np.int32)
        nm = np.zeros(n, np.int32)
        nk = np.zeros(n, np.int32)
        self._check(lib().rtvm_batch_status_at(self._h, back, C.byref(done), ptr(st), ptr(nm), ptr(nk), n))
        return done.value, st, nm, nk

    def batch_ticket(self):
        """Ticket of the batch queued last."""
        return int(lib().rtvm_batch_ticket(self._h))

    def batch_status_of(self, ticket, n):
        """(frames consumed, reason, status[n], matches[n], keypoints[n]) of the batch `ticket`
        (one of the last 16).  reason: PENDING_* when frames were left pending."""
        done, reason = C.c_int(0), C.c_int(0)
        st = np.zeros(n, np.int32)
        nm = np.zeros(n, np.int32)
        nk = np.zeros(n, np.int32)
        self._check(lib().rtvm_batch_status_of(self._h, ticket, C.byref(done), C.byref(reason), ptr(st), ptr(nm),
                                               ptr(nk), n))
        return done.value, reason.value, st, nm, nk

    def batch_cleared(self, ticket):
        """Index of the first frame of the piece / batch `ticket` whose blend cleared a canvas pixel, or -1."""
        first = C.c_int(-1)
        self._check(lib().rtvm_batch_cleared(self._h, ticket, C.byref(first)))
        return first.value

    def batch_retry(self, ticket, first, count):
        """Queues the canvas half again for frames first .. first+count-1 of the batch `ticket`;
        returns the ticket of the retry."""
        self._check(lib().rtvm_batch_retry(self._h, ticket, int(first), int(count)))
        return self.batch_ticket()

    def batch_detect(self, bgr_dev_ptrs):
        """Queues detection, matching and RANSAC of a batch (no canvas half); returns its ticket."""
        n = len(bgr_dev_ptrs)
        arr = (C.c_void_p * n)(*[C.c_void_p(p) for p in bgr_dev_ptrs])
        self._check(lib().rtvm_batch_detect(self._h, arr, n))
        return self.batch_ticket()

    def batch_release(self, ticket):
        self._check(lib().rtvm_batch_release(self._h, ticket))

    # ---- detected batches with the glue on the host ---------------------------------
    def _flag(self, rc):
        if rc < 0:
            self._check(rc)
        return bool(rc)

    def batch_results_ready(self, ticket):
        return self._flag(lib().rtvm_batch_results_ready(self._h, ticket))

    def batch_results(self, ticket, n):
        """(matches[n], found[n], keypoints[n], H_rel[n, 3, 3]) of the detected batch `ticket`."""
        nm = np.zeros(n, np.int32)
        fd = np.zeros(n, np.int32)
        nk = np.zeros(n, np.int32)
        H = np.zeros((n, 3, 3), np.float64)
        self._check(lib().rtvm_batch_results(self._h, ticket, ptr(nm), ptr(fd), ptr(nk), ptr(H), n))
        return nm, fd, nk, H

    def batch_commit(self, ticket, first, H_abs):
        """Queues warp + blend of frames first .. first+len(H_abs)-1 of the detected batch `ticket` with the
        host's absolute homographies; returns the ticket of the piece."""
        H = np.ascontiguousarray(H_abs, dtype=np.float64).reshape(-1, 9)
        self._check(lib().rtvm_batch_commit(self._h, ticket, int(first), len(H), ptr(H)))
        return self.batch_ticket()

    def batch_done(self, ticket):
        return self._flag(lib().rtvm_batch_done(self._h, ticket))

    def batch_rebase(self):
        self._check(lib().rtvm_batch_rebase(self._h))

    def batch_get_features(self, ticket, index):
        cap = max(self.max_keypoints(), 1)
        n = C.c_int(0)
        xy = np.empty((cap, 2), np.float32)
        size = np.empty(cap, np.float32)
        angle = np.empty(cap, np.float32)
        resp = np.empty(cap, np.float32)
        octave = np.empty(cap, np.int32)
        desc = self._desc_buffer(cap)
        self._check(lib().rtvm_batch_get_features(self._h, ticket, int(index), C.byref(n), ptr(xy), ptr(size), ptr(angle),
                                                  ptr(resp), ptr(octave), ptr(desc)))
        k = n.value
        return dict(xy=xy[:k].copy(), size=size[:k].copy(), angle=angle[:k].copy(),
                    response=resp[:k].copy(), octave=octave[:k].copy(), desc=desc[:k].copy())

    def batch_get_matches(self, ticket, index):
        cap = max(self.max_keypoints(), 1)
        m = C.c_int(0)
        qi = np.empty(cap, np.int32)
        ti = np.empty(cap, np.int32)
        di = np.empty(cap, np.int32)
        self._check(lib().rtvm_batch_get_matches(self._h, ticket, int(index), C.byref(m), ptr(qi), ptr(ti), ptr(di)))
        k = m.value
        return qi[:k].copy(), ti[:k].copy(), self._distances(di[:k].copy())

    def process_video(self, bgr_dev_ptrs, batch=16, trace=None, canvas_batch=None):
        """process_frame over a whole list of device-resident frames: rtvm_process_video.  Detection, matching and
        RANSAC run up to four batches ahead of the canvas half, which follows in pieces of canvas_batch frames
        (default: half a batch), PIECES_IN_FLIGHT in flight; frames that come back pending (behind a blend that
        cleared a canvas pixel, behind a frame the reference skips) are dealt with inside the library.
        bgr_dev_ptrs: a list of device addresses, or a frame source with acquire(first, k) -> addresses
        (HostFrameRing).  Returns the per-frame status codes; self.video_stats counts what was queued.  trace: a
        list that receives (seconds, first frame, length, frames consumed, reason) for every canvas half whose
        outcome was read."""
        n = len(bgr_dev_ptrs)
        status = np.zeros(max(n, 1), np.int32)
        stats = VideoStats()
        cap = 0 if trace is None else 8 * n + 64
        events = (VideoEvent * cap)() if cap else None
        n_events = C.c_int(0)
        acquire = getattr(bgr_dev_ptrs, "acquire", None)
        cb = int(canvas_batch or 0)
        if acquire is None and trace is None:
            arr = (C.c_void_p * max(n, 1))(*bgr_dev_ptrs)
            self._check(lib().rtvm_process_video(self._h, arr, n, int(batch), cb, ptr(status), C.byref(stats)))
        else:
            failure = []

            def source(user, first, count, dev):
                try:
                    got = acquire(first, count) if acquire else bgr_dev_ptrs[first:first + count]
                    for j in range(count):
                        dev[j] = got[j]
                    return OK
                except BaseException as e:   # must not unwind through the C frames
                    failure.append(e)
                    return E_CALLBACK
            fn = FRAME_SOURCE(source)
            # a frame source copies host frames while pieces run: one piece more keeps the device busy meanwhile
            in_flight = PIECES_IN_FLIGHT + 1 if acquire else PIECES_IN_FLIGHT
            rc = lib().rtvm_process_video_source(self._h, fn, None, n, int(batch), cb, in_flight, ptr(status), C.byref(stats),
                                                 events, cap, C.byref(n_events))
            if failure:
                raise failure[0]
            self._check(rc)
        if trace is not None:
            trace.extend((e.seconds, e.first, e.count, e.done, e.reason) for e in events[:min(cap, n_events.value)])
        self.video_stats = {k: getattr(stats, k) for k, _ in VideoStats._fields_}
        return status[:n]

    # ---- process_frame as a call that only takes the frame (rtvm_pipe_*) ----------------
    def pipe_open(self, batch, piece, glue_fn):
        """glue_fn: a GLUE_FN instance; the caller keeps it alive as long as the pipe."""
        self._check(lib().rtvm_pipe_open(self._h, int(batch), int(piece), glue_fn, None))

    def pipe_put(self, bgr):
        """bgr: C-contiguous u8 array of the context's frame shape (the caller has checked).  Returns the code."""
        return lib().rtvm_pipe_put(self._h, bgr.ctypes.data)   # the address as an integer: once per frame, no cast object

    def pipe_flush(self):
        return lib().rtvm_pipe_flush(self._h)

    def pipe_busy(self):
        return self._flag(lib().rtvm_pipe_busy(self._h))

    def pipe_last(self):
        """(ticket, index) of the newest frame in the batch that stays held after a flush, or None."""
        t, i = C.c_long(0), C.c_int(-1)
        self._check(lib().rtvm_pipe_last(self._h, C.byref(t), C.byref(i)))
        return (t.value, i.value) if t.value else None

    def pipe_stats(self):
        st = PipeStats()
        self._check(lib().rtvm_pipe_get_stats(self._h, C.byref(st)))
        return {k: getattr(st, k) for k, _ in PipeStats._fields_}

    def pipe_close(self):
        if getattr(self, "_h", None):
            self._check(lib().rtvm_pipe_close(self._h))

    # ---- frame ingest (host frames -> device ring) ------------------------------
    def ingest_reserve(self, n_slots):
        self._check(lib().rtvm_ingest_reserve(self._h, int(n_slots)))

    def ingest_write(self, slot, bgr):
        """Copies a host frame into the pinned slot (several threads for large frames); complete on return."""
        bgr = self._frame(bgr, "ingest_write")
        self._check(lib().rtvm_ingest_write(self._h, int(slot), C.c_void_p(bgr.ctypes.data)))

    def ingest_host_slot(self, slot):
        """The pinned host frame of a slot as a numpy array (frame_h, frame_w, 3) u8 to write into."""
        p = C.c_void_p()
        self._check(lib().rtvm_ingest_host_slot(self._h, int(slot), C.byref(p)))
        n = self.frame_h * self.frame_w * 3
        return np.ctypeslib.as_array((C.c_uint8 * n).from_address(p.value)).reshape(self.frame_h, self.frame_w, 3)

    def ingest_submit(self, slot):
        """Queues the copy of a slot to the device; returns the device address of the frame."""
        p = C.c_void_p()
        self._check(lib().rtvm_ingest_submit(self._h, int(slot), C.byref(p)))
        return p.value

    def ingest_submit_range(self, first_slot, count):
        """Queues one copy of `count` consecutive slots; returns their device addresses."""
        out = (C.c_void_p * int(count))()
        self._check(lib().rtvm_ingest_submit_range(self._h, int(first_slot), int(count), out))
        return [int(v) for v in out]

    def process_host_video(self, first_frame, frames, row_off, col_off, batch=16, trace=None):
        """The reference's loop (main.py:1596-1634) for frames in host memory: stream_begin with
        first_frame, then process_video over `frames` (a sequence of H x W x 3 u8 arrays), which
        travel through the ingest ring batch by batch while earlier batches are being stitched.
        Returns the per-frame status codes."""
        ring = getattr(self, "_ring", None)
        if ring is None:
            ring = self._ring = HostFrameRing(self, 5 * 32)
        ring.frames = []
        self.stream_begin(ring.upload_first(first_frame), row_off, col_off)
        ring.frames = frames
        try:
            return self.process_video(ring, batch, trace)
        finally:
            ring._settle()

    def frame_status(self):
        st, nk, nm = C.c_int(0), C.c_int(0), C.c_int(0)
        H = np.zeros(9, np.float64)
        self._check(lib().rtvm_frame_status(self._h, C.byref(st), C.byref(nk), C.byref(nm), ptr(H)))
        return st.value, nk.value, nm.value, H.reshape(3, 3)
