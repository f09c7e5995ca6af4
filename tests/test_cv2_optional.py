"""The CPU restatement against OpenCV itself.  OpenCV is not installed in the build image (every test here is
skipped there), so the restatement's parity with the library the reference calls is UNPINNED; these tests are what
pins it on a machine where `import cv2` works.  Stage by stage, on seeded synthetic inputs, with the call and
parameters of the reference's call site (main.py line in each test)."""
import numpy as np
import pytest

cv2 = pytest.importorskip("cv2")

import oracle_ffi as orc  # noqa: E402
import rtvm  # noqa: E402
from rtvm import synth  # noqa: E402


def frames(n=2, seed=0, shape=(240, 320)):
    return synth.make_video(shape[0], shape[1], n, world_size=(520, 620), seed=seed)[0]


def test_gray_main_py_717():
    f = frames(1)[0]
    assert np.array_equal(cv2.cvtColor(f, cv2.COLOR_BGR2GRAY), orc.gray(f))


def test_orb_keypoint_sets_and_descriptors_main_py_36_718():
    f = frames(1, seed=3)[0]
    kp, des = cv2.ORB_create(500).detectAndCompute(cv2.cvtColor(f, cv2.COLOR_BGR2GRAY), None)
    want = orc.orb(f, 500)
    # OpenCV leaves the order after retainBest unspecified: compare as sets, descriptors aligned by keypoint
    key = lambda x, y, o: (round(float(x), 3), round(float(y), 3), int(o))
    a = {key(k.pt[0], k.pt[1], k.octave): bytes(d) for k, d in zip(kp, des)}
    b = {key(x, y, o): bytes(d) for (x, y), o, d in zip(want["xy"], want["octave"], want["desc"])}
    assert set(a) == set(b)
    assert sum(a[k] == b[k] for k in a) >= 0.98 * len(a)     # fastAtan2's polynomial may flip a few descriptor bits
    ang = {key(k.pt[0], k.pt[1], k.octave): k.angle for k in kp}
    mine = {key(x, y, o): float(an) for (x, y), o, an in zip(want["xy"], want["octave"], want["angle"])}
    assert max(abs(ang[k] - mine[k]) for k in ang) < 0.5


def test_hamming_cross_check_main_py_695_698():
    rng = np.random.default_rng(1)
    t = rng.integers(0, 256, (700, 32), dtype=np.uint8)
    q = t[rng.integers(0, 700, 650)] ^ (rng.random((650, 32)) < 0.1).astype(np.uint8)
    ms = sorted(cv2.BFMatcher(cv2.NORM_HAMMING, crossCheck=True).match(q, t), key=lambda m: m.distance)
    qi, ti, di = orc.match_hamming(q, t, True)
    assert [(m.queryIdx, m.trainIdx, int(m.distance)) for m in ms] == list(zip(qi.tolist(), ti.tolist(), di.tolist()))


def test_l2_knn2_and_ratio_main_py_688_692():
    rng = np.random.default_rng(2)
    t = rng.integers(0, 120, (600, 128)).astype(np.float32)
    q = np.clip(t[rng.integers(0, 600, 500)] + np.rint(rng.normal(0, 8, (500, 128))), 0, 255).astype(np.float32)
    knn = cv2.BFMatcher().knnMatch(q, t, k=2)
    idx, dist = orc.match_l2_knn2(q, t)
    assert [[m.trainIdx for m in pair] for pair in knn] == idx.tolist()
    assert np.array_equal(np.array([[m.distance for m in pair] for pair in knn], np.float32), dist)


def test_find_homography_ransac_main_py_856():
    rng = np.random.default_rng(4)
    H = np.array([[1.01, 0.02, 5.0], [-0.015, 0.99, -3.0], [2e-5, -1e-5, 1.0]])
    src = rng.uniform(0, 320, (400, 2)).astype(np.float32)
    p = np.c_[src, np.ones(400)] @ H.T
    dst = (p[:, :2] / p[:, 2:]).astype(np.float32)
    dst[:80] += rng.uniform(-40, 40, (80, 2)).astype(np.float32)
    Hc, mask = cv2.findHomography(src.reshape(-1, 1, 2), dst.reshape(-1, 1, 2), cv2.RANSAC, ransacReprojThreshold=2.0)
    Ho, mo = orc.find_homography(src, dst, 2.0)
    assert np.array_equal(mask.ravel().astype(bool), mo.astype(bool))
    assert np.allclose(Hc, Ho, rtol=1e-6, atol=1e-8)


def test_warp_perspective_main_py_871():
    f = frames(1, seed=5)[0]
    H = np.array([[0.99, 0.03, 40.5], [-0.02, 1.01, 120.25], [1e-5, 2e-5, 1.0]])
    ch, cw = 480, 384
    ref = cv2.warpPerspective(f, H, (cw, ch), flags=cv2.INTER_LINEAR)
    d_float = np.abs(ref.astype(int) - orc.warp_perspective(f, H, ch, cw, 0).astype(int)).max()
    d_fixed = np.abs(ref.astype(int) - orc.warp_perspective(f, H, ch, cw, 1).astype(int)).max()
    assert min(d_float, d_fixed) == 0, (d_float, d_fixed)     # one of the two arithmetics is this OpenCV's


def test_distance_transform_main_py_888():
    rng = np.random.default_rng(6)
    mask = (rng.random((300, 400)) > 0.02).astype(np.uint8) * 255
    mask[100:180, 150:260] = 255
    assert np.array_equal(cv2.distanceTransform(mask, cv2.DIST_L2, 3), orc.distance_transform(mask))


def test_gaussian_blur_31_main_py_897():
    rng = np.random.default_rng(7)
    w = rng.random((200, 260)).astype(np.float32)
    assert np.abs(cv2.GaussianBlur(w, (31, 31), 0) - orc.gaussian_blur31(w)).max() <= 1e-6


def test_canny_main_py_489_1141():
    g = cv2.cvtColor(frames(1, seed=9)[0], cv2.COLOR_BGR2GRAY)
    assert np.array_equal(cv2.Canny(g, 50, 150), orc.canny(g, 50, 150))
    b = cv2.GaussianBlur(g, (3, 3), 0)
    assert np.array_equal(cv2.Canny(b, 40, 120), orc.canny(b, 40, 120))


def test_sift_keypoints_main_py_33():
    f = frames(1, seed=8)[0]
    kp, des = cv2.SIFT_create(700).detectAndCompute(cv2.cvtColor(f, cv2.COLOR_BGR2GRAY), None)
    want = orc.sift(f, 700)
    assert len(kp) == len(want["xy"])
    pts = np.array([k.pt for k in kp], np.float32)
    assert np.abs(pts - want["xy"]).max() < 0.05            # OpenCV's float exp / atan2 approximations vs libm
    assert np.abs(des - want["desc"]).max() <= 2
