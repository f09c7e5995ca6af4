"""CPU restatement of VideMosaic.process_frame (reference main.py:710-834) on top of the
oracle library.  Tests only.

glue="numpy": validate / smooth / chain exactly as the reference writes them
(np.linalg.det, np.linspace, np.matmul).  glue="explicit": the same arithmetic spelled
out operation by operation (2x2 LU determinant, sequential weighted sum, 3x3 product as
fma(a2, b2, fma(a1, b1, a0*b0)): the order of a BLAS dgemm kernel with fused multiply-add,
which is what np.matmul runs on such a CPU), which is what a device kernel reproduces bit for bit.
"""
import numpy as np

import oracle_ffi as orc

FEW, NO_H, INVALID = 1, 2, 4


class OracleMosaic:
    def __init__(self, first_image, output_height_times=2, output_width_times=1.2, nfeatures=700,
                 canvas_size=None, fixed=False, glue="numpy", detector="orb"):
        fh, fw = first_image.shape[:2]
        if canvas_size is None:
            canvas_size = (int(output_height_times * fh), int(output_width_times * fw))
        self.ch, self.cw = canvas_size
        self.nfeatures, self.fixed, self.glue = nfeatures, int(fixed), glue
        self.w_offset = int(self.ch / 1 - fh / 1)
        self.h_offset = int(self.cw / 2 - fw / 2)
        self.detector = detector
        self.feat_prev = self.detect(first_image)
        self.canvas = np.zeros((self.ch, self.cw, 3), np.uint8)
        self.canvas[self.w_offset:self.w_offset + fh, self.h_offset:self.h_offset + fw] = first_image
        self.H_old = np.eye(3)
        self.H_old[0, 2] = self.h_offset
        self.H_old[1, 2] = self.w_offset
        self.H = None
        self.history = []
        self.last_valid_H = np.eye(3)
        self.status = 0
        self.n_matches = 0

    # --- validate_homography, main.py:761-801
    def validate(self, H):
        if np.any(np.isnan(H)) or np.any(np.isinf(H)):
            return False
        if self.glue == "numpy":
            translation = np.sqrt(H[0, 2] ** 2 + H[1, 2] ** 2)
            with np.errstate(invalid="ignore"):
                scale = np.sqrt(np.linalg.det(H[:2, :2]))
        else:
            translation = np.sqrt(H[0, 2] * H[0, 2] + H[1, 2] * H[1, 2])
            a, b, c, d = H[0, 0], H[0, 1], H[1, 0], H[1, 1]
            if a == 0.0 and c == 0.0:
                det = 0.0   # zero pivot column: LAPACK's LU leaves a zero on the diagonal
            elif abs(a) >= abs(c):
                det = a * (d - (c * (1.0 / a)) * b)
            else:
                det = -(c * (b - (a * (1.0 / c)) * d))
            with np.errstate(invalid="ignore"):
                scale = np.sqrt(det)
        if translation > 50:
            return False
        if abs(scale - 1.0) > 0.3:
            return False
        if abs(H[2, 0]) > 0.001 or abs(H[2, 1]) > 0.001:
            return False
        return True

    # --- smooth_homography, main.py:803-834
    def smooth(self, H):
        self.history.append(H.copy())
        if len(self.history) > 5:
            self.history.pop(0)
        n = len(self.history)
        if n < 2:
            return H
        weights = np.linspace(0.5, 1.0, n)
        weights = weights / np.sum(weights)
        out = np.zeros_like(H)
        for i, h in enumerate(self.history):
            out += weights[i] * h
        return out

    def chain(self, A, B):
        if self.glue == "numpy":
            return np.matmul(A, B)
        return orc.chain3x3(A, B)

    def detect(self, frame):
        return orc.sift(frame, self.nfeatures) if self.detector == "sift" else orc.orb(frame, self.nfeatures)

    def match(self, cur, prev):
        if self.detector != "sift":
            return orc.match_hamming(cur["desc"], prev["desc"], True)
        # main.py:688-692, 698: knn-2, Lowe's ratio 0.7 (a float64 product in Python), stable sort by distance
        idx, dist = orc.match_l2_knn2(cur["desc"], prev["desc"])
        keep = dist[:, 0].astype(np.float64) < 0.7 * dist[:, 1].astype(np.float64)
        qi = np.nonzero(keep)[0]
        qi = qi[np.argsort(dist[qi, 0], kind="stable")]
        return qi, idx[qi, 0], dist[qi, 0]

    def process_frame(self, frame):
        cur = self.detect(frame)
        qi, ti, di = self.match(cur, self.feat_prev)
        self.feat_cur, self.matches = cur, (qi, ti, di)
        self.n_matches = len(qi)
        if len(qi) < 4:
            self.status = FEW
            return
        H_rel, _ = orc.find_homography(cur["xy"][qi], self.feat_prev["xy"][ti], 2.0)
        if H_rel is None:
            self.status = NO_H
            return
        self.status = 0
        if not self.validate(H_rel):
            H_rel = np.eye(3)
            self.status |= INVALID
        else:
            self.last_valid_H = H_rel.copy()
        H_s = self.smooth(H_rel)
        self.H = self.chain(self.H_old, H_s)
        self.canvas = orc.warp_blend(self.canvas, frame, self.H, self.fixed)
        self.H_old = self.H
        self.feat_prev = cur
