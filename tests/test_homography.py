"""findHomography(RANSAC): oracle known answers (CPU), HIP vs oracle (GPU)."""
import numpy as np
import pytest

import oracle_ffi as orc

HT = np.array([[1.01, 0.02, 3.5], [-0.015, 0.99, -2.25], [1e-5, -2e-5, 1.0]])


def make_pairs(seed, n, outlier_frac=0.0, noise=0.0, H=HT):
    rng = np.random.default_rng(seed)
    src = rng.uniform(0, 1280, (n, 2))
    src[:, 1] *= 720 / 1280
    src = src.astype(np.float32)
    p = np.c_[src.astype(np.float64), np.ones(n)] @ H.T
    dst = p[:, :2] / p[:, 2:]
    if noise:
        dst = dst + rng.normal(0, noise, dst.shape)
    dst = dst.astype(np.float32)
    k = int(n * outlier_frac)
    out = rng.permutation(n)[:k]
    dst[out] = rng.uniform(0, 700, (k, 2)).astype(np.float32)
    return src, dst, out


def proj(H, pts):
    p = np.c_[pts.astype(np.float64), np.ones(len(pts))] @ H.T
    return p[:, :2] / p[:, 2:]


# ------------------------------------------------------------------ CPU
def test_oracle_exact_data_recovers_h():
    src, dst, _ = make_pairs(0, 500)
    H, mask = orc.find_homography(src, dst)
    assert mask.all()
    assert np.abs(H - HT).max() < 5e-6
    assert H[2, 2] == 1.0


def test_oracle_outliers_rejected():
    src, dst, out = make_pairs(1, 600, 0.3, 0.3)
    H, mask = orc.find_homography(src, dst)
    assert mask[out].sum() <= 2
    good = np.setdiff1d(np.arange(600), out)
    assert mask[good].mean() > 0.9
    err = np.linalg.norm(proj(H, src[good]) - proj(HT, src[good]), axis=1)
    assert err.max() < 0.5


def test_oracle_mask_is_threshold_of_ransac_model():
    """every inlier is within the 2 px threshold of SOME homography close to the result."""
    src, dst, out = make_pairs(2, 300, 0.5, 0.5)
    H, mask = orc.find_homography(src, dst)
    e = np.linalg.norm(proj(H, src) - dst, axis=1)
    assert np.all(e[mask > 0] < 3.0)
    assert mask[out].sum() <= 2


def test_oracle_small_counts():
    src, dst, _ = make_pairs(3, 10)
    assert orc.find_homography(src[:3], dst[:3])[0] is None
    H, mask = orc.find_homography(src[:4], dst[:4])
    assert mask.all() and np.abs(H - HT).max() < 1e-3
    H, mask = orc.find_homography(src[:5], dst[:5])
    assert H is not None


def test_oracle_degenerate_returns_none():
    z = np.zeros((12, 2), np.float32)
    assert orc.find_homography(z, z)[0] is None
    line = np.c_[np.arange(20), 2 * np.arange(20)].astype(np.float32)
    assert orc.find_homography(line, line + 1)[0] is None


def test_oracle_rng_stream():
    """cv::RNG(-1): first outputs of the multiply-with-carry generator."""
    state = 0xFFFFFFFFFFFFFFFF
    outs = []
    for _ in range(3):
        state = ((state & 0xFFFFFFFF) * 4164903690 + (state >> 32)) & 0xFFFFFFFFFFFFFFFF
        outs.append(state & 0xFFFFFFFF)
    # 0xFFFFFFFF * 4164903690 + 0xFFFFFFFF = 0xF83F630A_07C09CF5 + 0xFFFFFFFF
    assert outs[0] == ((0xFFFFFFFF * 4164903690 + 0xFFFFFFFF) & 0xFFFFFFFF)
    assert len(set(outs)) == 3


# ------------------------------------------------------------------ GPU
GPU_CASES = [
    # (seed, n, outlier_frac, noise)
    (10, 5, 0.0, 0.0), (11, 8, 0.0, 0.2), (12, 64, 0.1, 0.3), (13, 65, 0.2, 0.3), (14, 500, 0.0, 0.0),
    (15, 700, 0.3, 0.4), (16, 1000, 0.15, 0.5), (17, 300, 0.6, 0.3), (18, 200, 0.75, 0.2), (19, 4000, 0.25, 0.5),
    (20, 40, 0.9, 0.2),
]


@pytest.mark.gpu
@pytest.mark.parametrize("seed,n,frac,noise", GPU_CASES)
def test_hip_ransac_matches_oracle(ctx_small, seed, n, frac, noise):
    src, dst, _ = make_pairs(seed, n, frac, noise)
    Hr, mr = orc.find_homography(src, dst)
    Hg, mg = ctx_small.find_homography(src, dst)
    assert (Hr is None) == (Hg is None)
    if Hr is None:
        return
    assert np.array_equal(mr, mg)
    assert np.array_equal(Hr, Hg), "max |dH| = %g" % np.abs(Hr - Hg).max()


@pytest.mark.gpu
def test_hip_ransac_small_and_degenerate(ctx_small):
    src, dst, _ = make_pairs(3, 10)
    assert ctx_small.find_homography(src[:3], dst[:3])[0] is None
    assert ctx_small.find_homography(src[:0], dst[:0])[0] is None
    Hg, mg = ctx_small.find_homography(src[:4], dst[:4])
    Hr, mr = orc.find_homography(src[:4], dst[:4])
    assert np.array_equal(Hg, Hr) and mg.all()
    z = np.zeros((12, 2), np.float32)
    assert ctx_small.find_homography(z, z)[0] is None
    dup = np.repeat(src[:2], 6, axis=0)
    Hr, _ = orc.find_homography(dup, dup)
    Hg, _ = ctx_small.find_homography(dup, dup)
    assert (Hr is None) == (Hg is None)


@pytest.mark.gpu
def test_hip_ransac_other_parameters(ctx_small):
    src, dst, _ = make_pairs(30, 400, 0.4, 0.6)
    for thresh, iters, conf in [(1.0, 2000, 0.995), (3.0, 50, 0.9), (2.0, 1, 0.5), (0.5, 500, 0.999)]:
        Hr, mr = orc.find_homography(src, dst, thresh, iters, conf)
        Hg, mg = ctx_small.find_homography(src, dst, thresh, iters, conf)
        assert (Hr is None) == (Hg is None)
        if Hr is not None:
            assert np.array_equal(mr, mg) and np.array_equal(Hr, Hg)
