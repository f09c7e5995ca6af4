"""numpy twins of parts of the CPU restatement.  Test infrastructure only.

Written independently of oracle/*.cpp (vectorised brute force instead of
loops) so that the two can check each other and generate the known-answer
fixtures under tests/golden/.  Parity unpinned (see oracle/oracle.h).
"""
import numpy as np

_POP = np.array([bin(i).count("1") for i in range(256)], np.int32)


def hamming_matrix(q, t):
    q = np.asarray(q, np.uint8).reshape(-1, 32)
    t = np.asarray(t, np.uint8).reshape(-1, 32)
    return _POP[q[:, None, :] ^ t[None, :, :]].sum(axis=2).astype(np.int32)


def match_hamming_xcheck(q, t, sorted_by_distance=False):
    """main.py:37,695(,698): mutual nearest neighbours, first minimum on ties."""
    q = np.asarray(q, np.uint8).reshape(-1, 32)
    t = np.asarray(t, np.uint8).reshape(-1, 32)
    if len(q) == 0 or len(t) == 0:
        z = np.zeros(0, np.int32)
        return z, z, z
    D = hamming_matrix(q, t)
    s = D.argmin(axis=1)           # numpy argmin returns the first minimum
    r = D.argmin(axis=0)
    qi = np.nonzero(r[s] == np.arange(len(q)))[0].astype(np.int32)
    ti = s[qi].astype(np.int32)
    di = D[qi, ti].astype(np.int32)
    if sorted_by_distance:
        o = np.argsort(di, kind="stable")
        qi, ti, di = qi[o], ti[o], di[o]
    return qi, ti, di


def match_l2_knn2(q, t):
    """main.py:34,688: two nearest by float32 L2, earlier index first on ties."""
    q = np.asarray(q, np.float32)
    t = np.asarray(t, np.float32)
    diff = q[:, None, :].astype(np.float32) - t[None, :, :]
    d2 = np.zeros(diff.shape[:2], np.float32)
    for k in range(diff.shape[2]):          # sequential float32 accumulation
        d2 = (d2 + diff[:, :, k] * diff[:, :, k]).astype(np.float32)
    d = np.sqrt(d2).astype(np.float32)
    o = np.argsort(d, axis=1, kind="stable")[:, :2]
    return o.astype(np.int32), np.take_along_axis(d, o, axis=1)


# ----------------------------------------------------------------- warp / blend
CHAMFER_A = 62587   # cvRound(0.955 * 65536)
CHAMFER_B = 89738   # cvRound(1.3693 * 65536)


def chamfer_dt_closed_form(mask):
    """main.py:888-889 as a closed form: the 3x3 chamfer distance to the nearest
    zero pixel is min over zero pixels of a*(max(|dx|,|dy|)-min(..)) + b*min(..)
    in Q16 (paths may cross any pixel; only the image rectangle bounds them, and
    the bounding box of two pixels is always inside it).  O(N * zeros): small masks only."""
    mask = np.asarray(mask)
    h, w = mask.shape
    zy, zx = np.nonzero(mask == 0)
    if len(zy) == 0:
        t = np.full((h, w), np.uint32(0xFFFFFFFF - CHAMFER_B), np.uint32)
        return t.astype(np.float32) * np.float32(1.0 / 65536.0)
    yy, xx = np.mgrid[0:h, 0:w]
    dx = np.abs(xx[:, :, None] - zx[None, None, :]).astype(np.int64)
    dy = np.abs(yy[:, :, None] - zy[None, None, :]).astype(np.int64)
    mn = np.minimum(dx, dy)
    mx = np.maximum(dx, dy)
    t = (CHAMFER_A * (mx - mn) + CHAMFER_B * mn).min(axis=2).astype(np.uint32)
    return t.astype(np.float32) * np.float32(1.0 / 65536.0)


def blend_numpy(output_img, warped_img, distance_transform, gaussian_blur31):
    """The numpy statements of main.py:878-927 executed as written, with the two
    OpenCV calls passed in as functions.  output_img: float64 canvas (as in the
    reference), warped_img: uint8.  Returns the new float64 canvas."""
    mask_new = np.any(warped_img > 0, axis=2).astype(np.uint8) * 255
    mask_old = np.any(output_img > 0, axis=2).astype(np.uint8) * 255
    overlap = mask_new & mask_old
    if np.any(overlap):
        dist_new = distance_transform(mask_new)
        dist_old = distance_transform(mask_old)
        dist_sum = dist_new + dist_old + np.float32(1e-6)
        weight_new = dist_new / dist_sum
        weight_old = dist_old / dist_sum
        weight_new = gaussian_blur31(weight_new.astype(np.float32))
        weight_old = gaussian_blur31(weight_old.astype(np.float32))
        weight_new_3ch = np.stack([weight_new, weight_new, weight_new], axis=2)
        weight_old_3ch = np.stack([weight_old, weight_old, weight_old], axis=2)
        blended = (output_img.astype(np.float32) * weight_old_3ch +
                   warped_img.astype(np.float32) * weight_new_3ch)
        overlap_3ch = np.stack([overlap, overlap, overlap], axis=2) > 0
        output_img = np.where(overlap_3ch, blended.astype(np.uint8), output_img)
        non_overlap_new = (mask_new & ~overlap) > 0
        non_overlap_3ch = np.stack([non_overlap_new] * 3, axis=2)
        output_img = np.where(non_overlap_3ch, warped_img, output_img)
    else:
        output_img = output_img.copy()
        output_img[warped_img > 0] = warped_img[warped_img > 0]
    return output_img
