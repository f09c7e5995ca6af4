"""VideMosaic: the reference's stitching class (main.py:15-977) on an MI355X.

Constructor and method signatures, attribute names, skip rules and printed
warnings follow the reference; the arithmetic runs in librtvm.so on the GPU
(there is no CPU fallback).  Per frame the class makes two library calls:
``frame_register`` (upload, ORB, match, RANSAC) and ``frame_commit`` (warp, blend,
state advance); the validate / smooth / chain arithmetic between them is the
reference's numpy, kept on the host so that the public thresholds, the history
list and H_old behave exactly like the reference's attributes.
"""
import threading

import numpy as np

from . import _ffi
from .types import KeyPoint, DMatch, keypoints_from_arrays, matches_from_arrays

_stage_local = threading.local()


def _stage_ctx(device=0):
    """Context without frame/canvas for the stand-alone matcher / RANSAC / finalisation calls:
    one per (thread, device), since a context serves one call at a time and lives on one GPU."""
    ctxs = getattr(_stage_local, "ctxs", None)
    if ctxs is None:
        ctxs = _stage_local.ctxs = {}
    if device not in ctxs:
        ctxs[device] = _ffi.Context(device=device)
    return ctxs[device]


class VideMosaic:
    def __init__(self, first_image, output_height_times=2, output_width_times=1.2, detector_type="sift",
                 show_intermediate=True, output_dir=None, visualize=True, *, nfeatures=700, canvas_size=None,
                 device=0, warp_mode="float"):
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
        det = _ffi.DET_ORB if detector_type == "orb" else _ffi.DET_SIFT_MATCH_ONLY
        self._ctx = _ffi.Context(fh, fw, self._ch, self._cw, det, nfeatures, flags, device)
        self._nfeatures = nfeatures
        self._cache = {}
        self._canvas_cache = None

        # main.py:86-87 (w_offset is the ROW offset, h_offset the COLUMN offset)
        self.w_offset = int(self._ch / 1 - fh / 1)
        self.h_offset = int(self._cw / 2 - fw / 2)
        self.process_first_frame(first_image)

        self.H_old = np.eye(3)          # main.py:92-94
        self.H_old[0, 2] = self.h_offset
        self.H_old[1, 2] = self.w_offset
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
        self.frame_prev = first_image
        self.frame_cur = first_image
        if self.detector_type == "sift":
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
        self.frame_cur = frame_cur
        if self.detector_type == "sift":
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

    def _finish_frame(self, frame_cur, n_matches, found, H_relative, staged=False):
        """main.py:722-759 after matching and RANSAC: skip rules, validate, smooth, chain, warp.
        staged: an ORB frame that was registered through the stage entry points, not frame_register."""
        if n_matches < 4:
            print(f"Предупреждение: Недостаточно совпадений ({n_matches}), пропуск кадра")
            return
        if not found:
            print("Предупреждение: Не удалось вычислить гомографию, пропуск кадра")
            return
        if not self.validate_homography(H_relative):
            print("Предупреждение: Невалидная гомография (тряска/размытие), использую последнюю валидную")
            H_relative = np.eye(3)
        else:
            self.last_valid_H = H_relative.copy()
        H_relative_smoothed = self.smooth_homography(H_relative)
        self.H = np.matmul(self.H_old, H_relative_smoothed)
        if self.detector_type == "sift":
            self._ctx.warp_blend(frame_cur, self.H)   # main.py:748
            self._sift_prev = self._sift_cur          # main.py:757-758
        elif staged:
            self._ctx.warp_blend(frame_cur, self.H)
        else:
            self._ctx.frame_commit(self.H)
        self._canvas_cache = None
        self.H_old = self.H
        self.frame_prev = self.frame_cur
        self._cache["advanced"] = True

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
        sift = self.detector_type == "sift"
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
            before = self.H_old
            self._finish_frame(frame, len(qi), H_relative is not None, H_relative, staged=True)
            if self.H_old is not before:   # the frame was stitched: it is the previous frame now
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
    def _features(self, slot):
        if self.detector_type == "sift":
            return self._sift_cur if slot == 0 else self._sift_prev
        key = "feat%d" % slot
        if key not in self._cache:
            self._cache[key] = self._ctx.get_features(slot)
        return self._cache[key]

    def _cur_slot(self):
        # after a committed frame the current features have become the previous ones
        return 0

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
        if self.detector_type == "sift":
            if "matches" not in self._cache:
                self._cache["matches"] = matches_from_arrays(*self._sift_matches)
            return self._cache["matches"]
        if "matches" not in self._cache:
            self._cache["matches"] = matches_from_arrays(*self._ctx.get_matches())
        return self._cache["matches"]

    @property
    def output_img(self):
        """The mosaic as the reference keeps it: (Hc, Wc, 3) float64 holding integers 0..255."""
        if self._canvas_cache is None:
            self._canvas_cache = self._ctx.canvas_read().astype(np.float64)
        return self._canvas_cache

    def output_img_u8(self):
        return self._ctx.canvas_read()

    def preview(self, width=400, height=300):
        """The whole mosaic scaled to width x height on the device (bilinear, as cv2.resize
        INTER_LINEAR), u8 BGR: what a progress display needs per frame (the reference's
        update_callback gets a copy of the full float64 canvas, main.py:1630-1632, which its GUI
        shrinks to 400 x 300, gui.py:143-155).  Only the small image crosses PCIe."""
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
        if np.any(np.isnan(H)) or np.any(np.isinf(H)):
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
        self.homography_history.append(H.copy())
        if len(self.homography_history) > self.history_size:
            self.homography_history.pop(0)
        n = len(self.homography_history)
        if n < 2:
            return H
        w = np.linspace(0.5, 1.0, n)
        w = w / np.sum(w)
        acc = np.zeros_like(H)
        for wi, hi in zip(w, self.homography_history):
            acc += wi * hi
        return acc

    def warp(self, frame_cur, H):
        """main.py:861-936: stitch frame_cur into the mosaic with homography H."""
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
