"""VideMosaic: the reference's stitching class (main.py:15-977) on an MI355X.

Constructor and method signatures, attribute names, skip rules and printed
warnings follow the reference; the arithmetic runs in librtvm.so on the GPU
(there is no CPU fallback).  The validate / smooth / chain arithmetic of
main.py:733-746 is the reference's numpy, kept on the host so that the public
thresholds, the history list and H_old behave exactly like the reference's
attributes.

``visualize`` and ``show_intermediate`` are accepted and change nothing: they only
open windows in the reference (main.py:703-707, 932-934) and no window is opened
here.  For every combination of them ``process_frame`` only takes the frame:
frames are detected, matched and fitted in batches on the device while the caller
fetches the next ones, the host applies the reference's rules to each batch's
results in frame order, and the warps and blends follow in pieces.  Reading any
result (``H``, ``H_old``, ``output_img``, ``matches``, ``kp_*``,
``homography_history``, ...) first brings the object to the state the reference
would have after the last ``process_frame`` call; warnings are printed in frame
order, at the latest then.  ``pipeline_batch=0`` opts out: every ``process_frame``
call then completes before it returns (``frame_register``: upload, detect, match,
RANSAC; ``frame_commit``: warp, blend, state advance).
"""
import ctypes
import threading

import numpy as np

from . import _ffi
from .types import KeyPoint, DMatch, keypoints_from_arrays, matches_from_arrays

_stage_local = threading.local()
_Mat9 = ctypes.c_double * 9


def _stage_ctx(device=0):
    """Context without frame/canvas for the stand-alone matcher / RANSAC / finalisation calls:
    one per (thread, device), since a context serves one call at a time and lives on one GPU."""
    ctxs = getattr(_stage_local, "ctxs", None)
    if ctxs is None:
        ctxs = _stage_local.ctxs = {}
    if device not in ctxs:
        ctxs[device] = _ffi.Context(device=device)
    return ctxs[device]


class _Pipeline:
    """process_frame without waiting (ORB or SIFT: the context's detector): the binding of rtvm_pipe_*.  Frames go
    through a ring of pinned host slots to the device; detection, matching and RANSAC run in batches of `batch`
    frames, up to four batches ahead; when a batch's results arrive the library calls back for each of its frames in
    order and the owner's own methods and attributes apply main.py:722-746 (skip rules, validate, smooth, chain);
    the frames to stitch are warped and blended in pieces of up to `piece` frames, three in flight.

    The homographies do not depend on the canvas, so a piece that comes back with frames pending (a blend cleared a
    canvas pixel: the canvas masks predicted for the frames behind it are wrong) is simply committed again from the
    first pending frame.  A frame the reference skips (main.py:722-731) invalidates the matching of everything
    detected behind it: those frames are detected again, matched against the last stitched frame.  All of that
    happens inside the library (csrc/video.hip); what stays here is the frame objects the rules are handed and the
    exception a rule may raise."""

    def __init__(self, vm, batch=32, piece=16):
        self.vm, self.ctx = vm, vm._ctx
        self.shape = (self.ctx.frame_h, self.ctx.frame_w, 3)
        self.frames = {}      # index -> the array handed to process_frame, until the rules have been applied to it
        self.n_in = 0
        self.error = None

        def glue(user, index, n_matches, found, H_rel, H_abs):
            try:
                frame = self.frames.pop(index)
                Hr = np.frombuffer(_Mat9.from_address(H_rel), np.float64).reshape(3, 3).copy()
                H = vm._glue(frame, n_matches, bool(found), Hr)
                if H is None:
                    return 0
                ctypes.memmove(H_abs, np.ascontiguousarray(H, np.float64).ctypes.data, 72)
                return 1
            except BaseException as e:   # must not unwind through the C frames; raised again by _check
                self.error = e
                return -1
        self._glue_fn = _ffi.GLUE_FN(glue)
        self.ctx.pipe_open(int(batch), max(1, min(int(piece), int(batch))), self._glue_fn)

    def _check(self, rc):
        if self.error is not None:
            e, self.error = self.error, None
            raise e
        self.ctx._check(rc)

    @property
    def stats(self):
        return self.ctx.pipe_stats()

    def put(self, frame):
        if frame.shape != self.shape:
            raise ValueError("process_frame: frame of shape %s, the mosaic was created for %s" % (frame.shape, self.shape))
        self.frames[self.n_in] = frame
        self.n_in += 1
        self._check(self.ctx.pipe_put(frame))   # complete on return: the caller may reuse the array

    def busy(self):
        return self.ctx.pipe_busy()

    def flush(self):
        self._check(self.ctx.pipe_flush())

    def last(self):
        return self.ctx.pipe_last()

    def close(self):
        self.frames.clear()
        self.ctx.pipe_close()


# np.linspace(0.5, 1.0, n) / sum for the history lengths seen so far (smooth_homography): the same values every
# frame, and a quarter of what the rules of a frame cost when computed anew.  The arrays are never written to.
_SMOOTH_WEIGHTS = {}


class VideMosaic:
    def __init__(self, first_image, output_height_times=2, output_width_times=1.2, detector_type="sift",
                 show_intermediate=True, output_dir=None, visualize=True, *, nfeatures=700, canvas_size=None,
                 device=0, warp_mode="float", pipeline_batch=32):
        # main.py:29-39
        self.detector_type = detector_type
        self.show_intermediate = show_intermediate
        self.output_dir = output_dir
        self.visualize = visualize
        if detector_type not in ("sift", "orb"):
            raise ValueError("detector_type must be 'sift' or 'orb'")
        if warp_mode not in ("float", "fixed"):
            raise ValueError("warp_mode must be 'float' or 'fixed'")
        # main.py:41-70: the YOLO models are outside this path; the reference itself
        # falls back to None when they cannot be loaded
        self.model = None
        self.model_world = None

        first_image = np.ascontiguousarray(first_image, dtype=np.uint8)
        fh, fw = first_image.shape[:2]
        if canvas_size is None:  # main.py:80-81
            canvas_size = (int(output_height_times * fh), int(output_width_times * fw))
        self._ch, self._cw = int(canvas_size[0]), int(canvas_size[1])
        flags = _ffi.FLAG_WARP_FIXED if warp_mode == "fixed" else 0
        # process_frame only enqueues (see the module docstring), whatever the display flags say: nothing observable
        # distinguishes the two modes once no window is opened; pipeline_batch = 0 turns it off
        self._deferred = pipeline_batch > 0 and nfeatures <= 7936
        # frame-synchronous SIFT runs stage by stage, the way the reference itself does
        self._staged_sift = detector_type == "sift" and not self._deferred
        det = _ffi.DET_ORB if detector_type == "orb" else \
            (_ffi.DET_SIFT_MATCH_ONLY if self._staged_sift else _ffi.DET_SIFT)
        self._ctx = _ffi.Context(fh, fw, self._ch, self._cw, det, nfeatures, flags, device)
        self._nfeatures = nfeatures
        self._cache = {}
        self._canvas_cache = None
        self._pipeline_batch = int(pipeline_batch)
        self._pipe = None
        self._pumping = False

        # main.py:86-87 (w_offset is the ROW offset, h_offset the COLUMN offset)
        self.w_offset = int(self._ch / 1 - fh / 1)
        self.h_offset = int(self._cw / 2 - fw / 2)
        self.process_first_frame(first_image)

        H0 = np.eye(3)                  # main.py:92-94
        H0[0, 2] = self.h_offset
        H0[1, 2] = self.w_offset
        self.H_old = H0
        self.H = None
        # main.py:97-102
        self.stabilization_enabled = True
        self.homography_history = []
        self.history_size = 5
        self.translation_threshold = 50
        self.scale_threshold = 0.3
        self.last_valid_H = np.eye(3)

    # ------------------------------------------------------------------ frames
    def process_first_frame(self, first_image):
        """main.py:104-112 plus the canvas set-up of main.py:80-90."""
        self._sync()
        if getattr(self, "_pipe", None) is not None:   # a new video: the stream's batches start over
            self._pipe.close()
            self._pipe = None
        self.frame_prev = first_image
        self.frame_cur = first_image
        if self._staged_sift:
            # stage by stage, the way the reference itself runs: features, canvas
            self._sift_prev = self._ctx.sift(first_image, self._nfeatures)
            self._sift_cur = self._sift_prev
            self._sift_matches = (np.zeros(0, np.int64), np.zeros(0, np.int32), np.zeros(0, np.float32))
            self._ctx.canvas_init(first_image, self.w_offset, self.h_offset)
        else:
            self._ctx.first_frame(first_image, self.w_offset, self.h_offset)
        self._cache.clear()
        self._canvas_cache = None

    def process_frame(self, frame_cur, frame_count):
        """main.py:710-759."""
        if self._deferred:
            if self._pipe is None:
                self._pipe = _Pipeline(self, self._pipeline_batch, max(1, self._pipeline_batch // 2))
            frame_cur = np.ascontiguousarray(frame_cur, dtype=np.uint8)
            self._frame_cur = frame_cur
            self._cache.clear()
            self._canvas_cache = None
            self._pumping = True
            try:
                self._pipe.put(frame_cur)
            finally:
                self._pumping = False
            return
        self.frame_cur = frame_cur
        if self._staged_sift:
            # main.py:716-731 with the SIFT branch of match(): detect, knn-2 + ratio, RANSAC
            cur = self._ctx.sift(frame_cur, self._nfeatures)
            registered = self._sift_register(self._ctx, cur, self._sift_prev)
            self._sift_finish(frame_cur, cur, registered)
        else:
            n_matches, found, H_relative = self._ctx.frame_register(frame_cur)
            self._cache.clear()
            self._finish_frame(frame_cur, n_matches, found, H_relative)

    @classmethod
    def _sift_register(cls, ctx, cur, prev):
        """main.py:721-731 for SIFT features: (matches as arrays, H_relative or None)."""
        if len(cur["xy"]) and len(prev["xy"]) >= 2:
            matches = cls._match_sift_arrays(cur["desc"], prev["desc"], ctx)
        else:
            matches = (np.zeros(0, np.int64), np.zeros(0, np.int32), np.zeros(0, np.float32))
        qi, ti, _ = matches
        H_relative = None
        if len(qi) >= 4:
            H_relative, _ = ctx.find_homography(cur["xy"][qi], prev["xy"][ti], 2.0)
        return matches, H_relative

    def _sift_finish(self, frame_cur, cur, registered):
        self._sift_cur = cur
        self._sift_matches, H_relative = registered
        self._cache.clear()
        self._finish_frame(frame_cur, len(self._sift_matches[0]), H_relative is not None, H_relative)

    def _glue(self, frame_cur, n_matches, found, H_relative):
        """main.py:722-746 and 756, 759 after matching and RANSAC: skip rules, validate, smooth, chain.
        Returns the absolute homography the frame is to be warped with, or None when the reference
        skips the frame (state not advanced)."""
        self._frame_cur = frame_cur
        if n_matches < 4:
            print(f"Предупреждение: Недостаточно совпадений ({n_matches}), пропуск кадра")
            return None
        if not found:
            print("Предупреждение: Не удалось вычислить гомографию, пропуск кадра")
            return None
        if not self.validate_homography(H_relative):
            print("Предупреждение: Невалидная гомография (тряска/размытие), использую последнюю валидную")
            H_relative = np.eye(3)
        else:
            self._last_valid_H = H_relative.copy()
        H_relative_smoothed = self.smooth_homography(H_relative)
        self._H = np.matmul(self._H_old, H_relative_smoothed)
        self._H_old = self._H
        self._frame_prev = frame_cur
        return self._H

    def _finish_frame(self, frame_cur, n_matches, found, H_relative, staged=False):
        """main.py:722-759 after matching and RANSAC, frame by frame: the rules above, then warp + blend.
        staged: an ORB frame that was registered through the stage entry points, not frame_register."""
        H = self._glue(frame_cur, n_matches, found, H_relative)
        if H is None:
            return
        if self._staged_sift:
            self._ctx.warp_blend(frame_cur, H)        # main.py:748
            self._sift_prev = self._sift_cur          # main.py:757-758
        elif staged:
            self._ctx.warp_blend(frame_cur, H)
        else:
            self._ctx.frame_commit(H)
        self._canvas_cache = None
        self._cache["advanced"] = True

    # ------------------------------------------------------------------ deferred work
    def _sync(self):
        """Brings the object to the state the reference has after the last process_frame call."""
        p = self._pipe
        if p is None or self._pumping or not p.busy():
            return
        self._pumping = True
        try:
            p.flush()
        finally:
            self._pumping = False
        self._cache.clear()
        self._canvas_cache = None

    def flush(self):
        """Waits until every frame handed to process_frame has been dealt with (a no-op with
        pipeline_batch=0).  Reading a result does the same."""
        self._sync()

    def _synced(name):   # noqa: N805 - attribute whose value depends on the frames still in flight
        def get(self):
            self._sync()
            return getattr(self, name)

        def put(self, value):
            self._sync()
            setattr(self, name, value)
        return property(get, put)

    H = _synced("_H")
    H_old = _synced("_H_old")
    last_valid_H = _synced("_last_valid_H")
    homography_history = _synced("_homography_history")
    frame_prev = _synced("_frame_prev")
    frame_cur = _synced("_frame_cur")
    # the knobs of main.py:97-101: setting one applies to the frames handed in afterwards, as in the reference
    stabilization_enabled = _synced("_stabilization_enabled")
    history_size = _synced("_history_size")
    translation_threshold = _synced("_translation_threshold")
    scale_threshold = _synced("_scale_threshold")
    del _synced

    def process_frames(self, frames, first_frame_count=1, workers=4):
        """process_frame(frames[0], first_frame_count), process_frame(frames[1], ...), ... with the
        same results, state and printed warnings, for callers that have several frames at hand
        (a decoder's queue).  Detection, matching and RANSAC of a frame depend only on the frame
        and its predecessor, so they run on `workers` threads, each with a device context of its
        own, ahead of the sequential part (skip rules, validate, smooth, chain, warp + blend), which
        keeps frame order.  Every frame is matched against its predecessor on the assumption that
        the predecessor is not skipped; after a skipped frame (main.py:722-731) the next one is
        registered again against the frame the reference would have used.  With ORB the last frame
        goes through process_frame itself, after the previous features of the device stream have
        been set to those of the last stitched frame, so that the object is left exactly as by
        per-frame calls.  (Frames already on the GPU: Context.process_video is the faster path.)"""
        frames = list(frames)
        if self._deferred:   # process_frame itself does not wait
            for i, frame in enumerate(frames):
                self.process_frame(frame, first_frame_count + i)
            return
        sift = self._staged_sift
        n = len(frames) if sift else len(frames) - 1   # ORB: the last frame takes the ordinary path
        if workers < 2 or n < 2:
            for i, frame in enumerate(frames):
                self.process_frame(frame, first_frame_count + i)
            return
        import gc
        import queue
        from concurrent.futures import Future, ThreadPoolExecutor

        fh, fw = self._ctx.frame_h, self._ctx.frame_w
        if getattr(self, "_worker_ctxs", None) is None or len(self._worker_ctxs) < workers:
            have = getattr(self, "_worker_ctxs", None) or []
            make = (lambda: _ffi.Context(device=self._ctx.device)) if sift else \
                (lambda: _ffi.Context(fh, fw, 0, 0, _ffi.DET_ORB, self._nfeatures, _ffi.FLAG_SINGLE_STREAM, self._ctx.device))
            self._worker_ctxs = have + [make() for _ in range(workers - len(have))]
        free = queue.SimpleQueue()
        for c in self._worker_ctxs[:workers]:
            free.put(c)
        detected = [Future() for _ in range(n)]
        nfeatures = self._nfeatures
        if sift:
            prev0 = self._sift_prev
            detect = lambda ctx, frame: ctx.sift(frame, nfeatures)
            register = self._sift_register
        else:
            prev0 = self._ctx.get_features(1)
            detect = lambda ctx, frame: ctx.orb(frame)
            register = self._orb_register
        state = {"prev": prev0, "prev_frame": None}

        def work(t):
            ctx = free.get()
            try:
                try:
                    cur = detect(ctx, np.ascontiguousarray(frames[t], dtype=np.uint8))
                except BaseException as e:
                    detected[t].set_exception(e)
                    raise
                detected[t].set_result(cur)
                # the task of frame t-1 started before this one and waits only for earlier frames
                prev = prev0 if t == 0 else detected[t - 1].result()
                return cur, prev, register(ctx, cur, prev)
            finally:
                free.put(ctx)

        def finish(frame, cur, registered):
            if sift:
                self._sift_finish(frame, cur, registered)
                return
            # the ORB frame, stitched through the stage entry points: same arithmetic as
            # frame_register / frame_commit, the previous features kept here instead of on the device
            (qi, ti, di), H_relative = registered
            self._cache.clear()
            before = self._H_old
            self._finish_frame(frame, len(qi), H_relative is not None, H_relative, staged=True)
            if self._H_old is not before:   # the frame was stitched: it is the previous frame now
                state["prev"], state["prev_frame"] = cur, frame

        ahead = 4 * workers
        # a full pass of the cyclic garbage collector stops every thread for tens of milliseconds in a
        # process that has imported numpy and torch; what this loop allocates is freed by reference counts
        collector_was_on = gc.isenabled()
        gc.disable()
        try:
            with ThreadPoolExecutor(max_workers=workers) as pool:
                tasks = [pool.submit(work, t) for t in range(min(ahead, n))]
                for t in range(n):
                    cur, prev, registered = tasks[t].result()
                    tasks[t] = None
                    if t + ahead < n:
                        tasks.append(pool.submit(work, t + ahead))
                    self.frame_cur = frames[t]
                    actual_prev = self._sift_prev if sift else state["prev"]
                    if prev is not actual_prev:   # the predecessor was skipped
                        registered = register(self._ctx, cur, actual_prev)
                    finish(frames[t], cur, registered)
        finally:
            if collector_was_on:
                gc.enable()
        if not sift:
            if state["prev_frame"] is not None:
                self._ctx.stream_set_prev(state["prev_frame"])
            self.process_frame(frames[-1], first_frame_count + len(frames) - 1)

    @staticmethod
    def _orb_register(ctx, cur, prev):
        """main.py:721-731 for ORB features: (matches as arrays sorted by distance, H_relative or None)."""
        if len(cur["xy"]) and len(prev["xy"]):
            matches = ctx.match_hamming(cur["desc"], prev["desc"], sorted_by_distance=True)
        else:
            matches = (np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(0, np.int32))
        qi, ti, _ = matches
        H_relative = None
        if len(qi) >= 4:
            H_relative, _ = ctx.find_homography(cur["xy"][qi], prev["xy"][ti], 2.0)
        return matches, H_relative

    # ------------------------------------------------------------------ lazily fetched state
    def _held_frame(self):
        """(ticket, index) of the newest frame when it went through the pipeline, else None."""
        return None if self._pipe is None else self._pipe.last()

    def _features(self, slot):
        if self._staged_sift:
            return self._sift_cur if slot == 0 else self._sift_prev
        self._sync()
        key = "feat%d" % slot
        if key not in self._cache:
            held = self._held_frame() if slot == 0 else None
            self._cache[key] = self._ctx.batch_get_features(*held) if held else self._ctx.get_features(slot)
        return self._cache[key]

    @property
    def kp_cur(self):
        return keypoints_from_arrays(self._features(0))

    @property
    def des_cur(self):
        return self._features(0)["desc"]

    @property
    def kp_prev(self):
        return keypoints_from_arrays(self._features(1))

    @property
    def des_prev(self):
        return self._features(1)["desc"]

    @property
    def matches(self):
        if self._staged_sift:
            if "matches" not in self._cache:
                self._cache["matches"] = matches_from_arrays(*self._sift_matches)
            return self._cache["matches"]
        self._sync()
        if "matches" not in self._cache:
            held = self._held_frame()
            arrays = self._ctx.batch_get_matches(*held) if held else self._ctx.get_matches()
            self._cache["matches"] = matches_from_arrays(*arrays)
        return self._cache["matches"]

    @property
    def output_img(self):
        """The mosaic as the reference keeps it: (Hc, Wc, 3) float64 holding integers 0..255."""
        self._sync()
        if self._canvas_cache is None:
            self._canvas_cache = self._ctx.canvas_read_f64()
        return self._canvas_cache

    def output_img_u8(self):
        self._sync()
        return self._ctx.canvas_read()

    def preview(self, width=400, height=300):
        """The whole mosaic scaled to width x height on the device (bilinear, as cv2.resize
        INTER_LINEAR), u8 BGR: what a progress display needs per frame (the reference's
        update_callback gets a copy of the full float64 canvas, main.py:1630-1632, which its GUI
        shrinks to 400 x 300, gui.py:143-155).  Only the small image crosses PCIe."""
        self._sync()
        ch, cw = self._ctx.canvas_h, self._ctx.canvas_w
        return self._ctx.canvas_read_scaled(0, 0, cw, ch, int(width), int(height))

    # ------------------------------------------------------------------ stages
    def match(self, des_cur, des_prev):
        """main.py:676-708: matches sorted by distance."""
        if self.detector_type == "sift":
            return matches_from_arrays(*self._match_sift_arrays(des_cur, des_prev, self._ctx))
        qi, ti, di = self._ctx.match_hamming(des_cur, des_prev, sorted_by_distance=True)
        return matches_from_arrays(qi, ti, di)

    @staticmethod
    def _match_sift_arrays(des_cur, des_prev, ctx=None):
        """main.py:688-698: knn-2, ratio test at 0.7, sorted by distance; (queryIdx, trainIdx, distance) arrays."""
        des_cur = np.asarray(des_cur, np.float32)
        des_prev = np.asarray(des_prev, np.float32)
        if len(des_prev) < 2:
            raise ValueError("not enough values to unpack (expected 2, got %d)" % len(des_prev))
        idx, dist = (ctx or _stage_ctx()).match_l2_knn2(des_cur, des_prev)
        keep = dist[:, 0].astype(np.float64) < 0.7 * dist[:, 1].astype(np.float64)
        qi = np.nonzero(keep)[0]
        qi = qi[np.argsort(dist[qi, 0], kind="stable")]
        return qi, idx[qi, 0], dist[qi, 0]

    @staticmethod
    def findHomography(image_1_kp, image_2_kp, matches):
        """main.py:836-859: RANSAC homography image 1 -> image 2, or None."""
        n = len(matches)
        p1 = np.zeros((n, 2), np.float32)
        p2 = np.zeros((n, 2), np.float32)
        for i, m in enumerate(matches):
            p1[i] = image_1_kp[m.queryIdx].pt
            p2[i] = image_2_kp[m.trainIdx].pt
        H, _ = _stage_ctx().find_homography(p1, p2, 2.0)
        return H

    def validate_homography(self, H):
        """main.py:761-801."""
        if H is None:
            return False
        if not np.isfinite(H).all():            # np.any(np.isnan(H)) or np.any(np.isinf(H)), in one pass
            return False
        shift = np.sqrt(H[0, 2] ** 2 + H[1, 2] ** 2)
        zoom = np.sqrt(np.linalg.det(H[:2, :2]))
        if shift > self.translation_threshold:
            print(f"Предупреждение: Обнаружено большое смещение ({shift:.1f}px), возможна тряска")
            return False
        if abs(zoom - 1.0) > self.scale_threshold:
            print(f"Предупреждение: Обнаружено большое изменение масштаба ({zoom:.2f}), возможна тряска")
            return False
        if abs(H[2, 0]) > 0.001 or abs(H[2, 1]) > 0.001:
            print("Предупреждение: Обнаружены сильные перспективные искажения")
            return False
        return True

    def smooth_homography(self, H):
        """main.py:803-834: recency-weighted mean of the last history_size relative homographies."""
        if not self.stabilization_enabled:
            return H
        history = self.homography_history
        history.append(H.copy())
        if len(history) > self.history_size:
            history.pop(0)
        n = len(history)
        if n < 2:
            return H
        w = _SMOOTH_WEIGHTS.get(n)
        if w is None:
            w = np.linspace(0.5, 1.0, n)
            w = _SMOOTH_WEIGHTS[n] = w / np.sum(w)
        acc = np.zeros_like(H)
        for wi, hi in zip(w, history):
            acc += wi * hi
        return acc

    def warp(self, frame_cur, H):
        """main.py:861-936: stitch frame_cur into the mosaic with homography H."""
        self._sync()
        self._ctx.warp_blend(frame_cur, H)
        self._canvas_cache = None
        return self.output_img

    @staticmethod
    def get_transformed_corners(frame_cur, H):
        """main.py:938-962: the frame's corners under H, truncated to int32, shape (1, 4, 2)."""
        h, w = frame_cur.shape[:2]
        pts = np.array([[0, 0], [w, 0], [w, h], [0, h]], np.float32).astype(np.float64)
        Hd = np.asarray(H, np.float64)
        d = pts[:, 0] * Hd[2, 0] + pts[:, 1] * Hd[2, 1] + Hd[2, 2]
        out = np.zeros((4, 2), np.float32)
        ok = np.abs(d) > np.finfo(np.float32).eps
        inv = np.where(ok, 1.0 / np.where(ok, d, 1.0), 0.0)
        out[:, 0] = ((pts[:, 0] * Hd[0, 0] + pts[:, 1] * Hd[0, 1] + Hd[0, 2]) * inv).astype(np.float32)
        out[:, 1] = ((pts[:, 0] * Hd[1, 0] + pts[:, 1] * Hd[1, 1] + Hd[1, 2]) * inv).astype(np.float32)
        return np.array(out[None], dtype=np.int32)

    def draw_border(self, image, corners, color=(0, 0, 0)):
        """main.py:964-977: 5-pixel polyline through the corners (display aid only)."""
        pts = corners[0]
        for i in range(len(pts) - 1, -1, -1):
            _thick_line(image, pts[i], pts[i - 1], color, 5)
        return image

    # ------------------------------------------------------------------ out of this path
    def detect_people(self, frame):
        return []

    def detect_objects(self, frame):
        return []

    def close(self):
        if self._pipe is not None:
            self._pipe.close()
            self._pipe = None
        for c in getattr(self, "_worker_ctxs", None) or []:
            c.close()
        self._worker_ctxs = None
        self._ctx.close()


def crop_black_areas(image, threshold=15, margin=5):
    """main.py:980-1003: cut the black surroundings (and a margin) off a mosaic.  `image` is a
    numpy image, or a VideMosaic: then the rectangle is found on the device canvas and only the
    rectangle crosses PCIe before the canvas is read."""
    if isinstance(image, VideMosaic):
        canvas = image.output_img_u8()
        rect = image._ctx.canvas_crop_rect(threshold)
        image = canvas
    else:
        if image.dtype != np.uint8:
            image = image.astype(np.uint8)
        rect = _stage_ctx().crop_rect(image, threshold)
    if rect is None:
        return image
    x, y, w, h = rect
    x = max(0, x + margin)
    y = max(0, y + margin)
    w = min(image.shape[1] - x, w - 2 * margin)
    h = min(image.shape[0] - y, h - 2 * margin)
    return image[y:y + h, x:x + w]


def scale_to_screen(image, target_w=None, target_h=None):
    """main.py:1006-1038: resize to fit the screen (1920 x 1080 when no size is given: the
    reference asks Windows and falls back to that), aspect ratio kept."""
    ih, iw = image.shape[0], image.shape[1]
    screen_w, screen_h = target_w, target_h
    if screen_w is None or screen_h is None:
        screen_w, screen_h = 1920, 1080
    scale = min(screen_w / float(iw), screen_h / float(ih))
    if scale <= 0:
        scale = 1.0
    new_w = max(1, int(iw * scale))
    new_h = max(1, int(ih * scale))
    if image.dtype != np.uint8:
        image = image.astype(np.uint8)
    return _stage_ctx().resize_linear(image, new_w, new_h)


def _thick_line(image, p0, p1, color, thickness):
    x0, y0, x1, y1 = int(p0[0]), int(p0[1]), int(p1[0]), int(p1[1])
    n = max(abs(x1 - x0), abs(y1 - y0), 1)
    r = thickness // 2
    h, w = image.shape[:2]
    for t in range(n + 1):
        x = x0 + (x1 - x0) * t // n
        y = y0 + (y1 - y0) * t // n
        ya, yb, xa, xb = max(0, y - r), min(h, y + r + 1), max(0, x - r), min(w, x + r + 1)
        if ya < yb and xa < xb:
            image[ya:yb, xa:xb] = color
