"""gray + ORB: oracle sanity (CPU), HIP vs oracle (GPU, every output bit-exact)."""
import numpy as np
import pytest

import oracle_ffi as orc
import rtvm
from rtvm import synth


def frame(h, w, seed=0, t=0):
    world = synth.make_world(h + 160, w + 160, seed=seed)
    G = synth.camera_matrix(t, 40.0, 60.0, (0.3, -0.2), step=3.0, rot_deg=2.0, zoom=1e-3)
    return synth.render_frame(world, G, h, w)


# ------------------------------------------------------------------ CPU
def test_gray_formula():
    rng = np.random.default_rng(0)
    f = rng.integers(0, 256, (37, 53, 3)).astype(np.uint8)
    ref = ((f[..., 0].astype(np.int64) * 3735 + f[..., 1].astype(np.int64) * 19235 +
            f[..., 2].astype(np.int64) * 9798 + 16384) >> 15).astype(np.uint8)
    assert np.array_equal(orc.gray(f), ref)
    assert orc.gray(np.full((4, 4, 3), 255, np.uint8)).min() == 255


def test_pyramid_geometry_and_resize():
    g = orc.gray(frame(240, 320))
    pyr, blur = orc.orb_pyramid(g)
    assert [p.shape for p in pyr][:3] == [(240, 320), (200, 267), (167, 222)]
    assert np.array_equal(pyr[0], g)
    # a constant image stays constant through resize and blur (kernel taps sum to 257/256)
    cpyr, cblur = orc.orb_pyramid(np.full((120, 160), 100, np.uint8))
    assert all((p == 100).all() for p in cpyr)
    assert all((b == 101).all() for b in cblur)
    # bilinear down-scaling stays within the range of the source
    assert pyr[1].min() >= g.min() and pyr[1].max() <= g.max()


def test_oracle_orb_invariants():
    f = frame(240, 320)
    r = orc.orb(f, 500)
    n = len(r["xy"])
    assert 300 < n <= 500
    counts = np.bincount(r["octave"], minlength=8)
    quota, nd, f = [], None, np.float32(1.0 / float(np.float32(1.2)))
    nd = np.float32(500) * (np.float32(1) - f) / (np.float32(1) - np.float32(float(f) ** 8))
    for _ in range(7):
        quota.append(int(np.rint(nd)))
        nd = np.float32(nd * f)
    quota.append(max(500 - sum(quota), 0))
    assert all(c <= q for c, q in zip(counts, quota))
    assert np.all(np.diff(r["octave"]) >= 0)
    scale = (np.float32(1.2) ** np.arange(8)).astype(np.float32)
    lev = r["xy"] / scale[r["octave"]][:, None]
    assert np.abs(lev - np.rint(lev)).max() < 1e-3          # integer level coordinates
    lw = np.array([320, 267, 222, 185, 154, 129, 107, 89])[r["octave"]]
    assert np.all(np.rint(lev[:, 0]) >= 31) and np.all(np.rint(lev[:, 0]) < lw - 31)
    assert np.all((r["angle"] >= 0) & (r["angle"] < 360.0001))
    assert np.allclose(r["size"], 31 * scale[r["octave"]])
    # descriptors of the same scene under a 3 px shift mostly match
    r2 = orc.orb(frame(240, 320, t=1), 500)
    qi, ti, di = orc.match_hamming(r2["desc"], r["desc"], True)
    assert len(qi) > 150 and np.median(di) < 40


def test_fast_detects_a_synthetic_corner():
    f = np.full((100, 100, 3), 40, np.uint8)
    yy, xx = np.mgrid[0:100, 0:100]
    shade = (250 - 2 * np.hypot(yy - 40, xx - 40)).clip(120, 250).astype(np.uint8)   # unique maximum at the tip
    f[40:, 40:] = shade[40:, 40:, None]
    r = orc.orb(f, 50)
    lvl0 = r["xy"][r["octave"] == 0]
    assert len(lvl0) >= 1
    assert np.abs(lvl0 - np.array([40, 40])).max(axis=1).min() <= 3


# ------------------------------------------------------------------ GPU
def _check_equal(got, ref):
    assert len(got["xy"]) == len(ref["xy"])
    for k in ("octave", "xy", "size", "response", "angle", "desc"):
        assert np.array_equal(got[k], ref[k]), k


@pytest.mark.gpu
@pytest.mark.parametrize("h,w,n", [(240, 320, 500), (200, 200, 50), (480, 640, 700), (720, 1280, 1000),
                                   # widths and heights around the tiles of the chained resize: not multiples of 4, primes
                                   (251, 331, 300), (333, 509, 400), (257, 258, 200), (130, 1027, 300)])
def test_hip_orb_bit_exact(h, w, n):
    c = rtvm.Context(h, w, 0, 0, nfeatures=n)
    for t in (0, 1):
        f = frame(h, w, seed=3, t=t)
        assert np.array_equal(c.gray(f), orc.gray(f))
        _check_equal(c.orb(f), orc.orb(f, n))
    c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("h,w,n", [(1080, 1920, 1000), (2160, 3840, 4000)])
def test_hip_orb_bit_exact_at_the_1080p_and_4k_sizes(h, w, n):
    """The detection sizes of the 1080p streams and of the 4K / 4000-feature configuration: tile runs of the FAST
    kernel over far more tiles, the selection's histogram and rank counting at 4000 features, the chained resize's
    rectangle tables of a large pyramid.  One textured frame each, all six outputs."""
    world = synth.make_world(h + 96, w + 96, seed=8, n_shapes=6000)
    f = np.ascontiguousarray(world[48:48 + h, 48:48 + w])
    c = rtvm.Context(h, w, 0, 0, nfeatures=n)
    got, ref = c.orb(f), orc.orb(f, n)
    assert np.array_equal(c.gray(f), orc.gray(f))
    assert len(ref["xy"]) >= n * 0.95 and np.bincount(ref["octave"], minlength=8).min() > 0
    _check_equal(got, ref)
    c.close()


@pytest.mark.gpu
def test_hip_orb_frames_too_small_for_a_pyramid():
    """Frames of fewer than 29 pixels a side: their upper levels are too small for the resize kernels and no level
    can hold a keypoint; gray still works and the feature set is empty, as the oracle's."""
    for h, w in ((20, 400), (300, 24)):
        rng = np.random.default_rng(h)
        f = rng.integers(0, 256, (h, w, 3)).astype(np.uint8)
        c = rtvm.Context(h, w, 0, 0, nfeatures=100)
        assert np.array_equal(c.gray(f), orc.gray(f))
        assert len(c.orb(f)["xy"]) == 0 and len(orc.orb(f, 100)["xy"]) == 0
        c.close()


@pytest.mark.gpu
def test_hip_orb_featureless_and_tiny():
    c = rtvm.Context(120, 160, 0, 0, nfeatures=100)
    flat = np.full((120, 160, 3), 77, np.uint8)
    assert len(c.orb(flat)["xy"]) == 0 and len(orc.orb(flat, 100)["xy"]) == 0
    f = frame(120, 160, seed=9)
    _check_equal(c.orb(f), orc.orb(f, 100))
    c.close()
    c = rtvm.Context(64, 64, 0, 0, nfeatures=100)   # only level 0 is larger than the 31 px border
    f = frame(64, 64, seed=4)
    _check_equal(c.orb(f), orc.orb(f, 100))
    c.close()


@pytest.mark.gpu
def test_hip_orb_many_ties():
    """A periodic pattern gives many equal FAST scores and Harris responses at the cuts."""
    f = np.full((256, 256, 3), 30, np.uint8)
    for y in range(8, 256, 24):
        for x in range(8, 256, 24):
            f[y:y + 10, x:x + 10] = 220
    c = rtvm.Context(256, 256, 0, 0, nfeatures=60)
    got, ref = c.orb(f), orc.orb(f, 60)
    _check_equal(got, ref)
    c.close()


@pytest.mark.gpu
def test_hip_legacy_gray_coefficients_match_oracle():
    """RTVM_FLAG_GRAY_LEGACY: cvtColor(BGR2GRAY) with the 14-bit coefficients of OpenCV <= 3.x, and the ORB features
    detected on that gray image, against the oracle in the same mode."""
    import rtvm
    frames, _ = rtvm.synth.make_video(240, 320, 1, world_size=(520, 620), seed=11)
    f = frames[0]
    ctx = rtvm.Context(240, 320, 480, 384, nfeatures=500, flags=rtvm._ffi.FLAG_GRAY_LEGACY)
    plain = rtvm.Context(240, 320, 480, 384, nfeatures=500)
    before = orc.set_gray_legacy(True)
    try:
        want_gray, want = orc.gray(f), orc.orb(f, 500)
    finally:
        orc.set_gray_legacy(before)
    got_gray, got = ctx.gray(f), ctx.orb(f)
    assert np.array_equal(got_gray, want_gray)
    diff = got_gray.astype(int) - plain.gray(f).astype(int)
    assert np.abs(diff).max() == 1 and 0 < (diff != 0).mean() < 0.5      # the two conversions differ by one level here and there
    for k in ("xy", "size", "angle", "response", "octave", "desc"):
        assert np.array_equal(got[k], want[k]), k
    ctx.close()
    plain.close()
