"""cv2.Canny as the reference calls it (main.py:489 `Canny(blurred, 40, 120)` and main.py:1141
`Canny(gray, 50, 150)`: aperture 3, L1 gradient).  CPU tests pin the oracle's restatement against
known answers and against a numpy statement of the same rules built independently (correlation
with the Sobel kernels, sector classification with tangents in floating point checked away from
the sector borders, connected components by scipy.ndimage.label); GPU tests compare the HIP path
with the oracle bit for bit through the C ABI.  Parity with OpenCV itself is unpinned (cv2 is not
importable here); tests/test_cv2_optional.py holds the comparison for the day it is."""
import importlib

import numpy as np
import pytest
from scipy import ndimage

import oracle_ffi as orc

rtvm = importlib.import_module("real-time-video-mosaic_amd")


def picture(h, w, seed):
    """Shapes with sharp and soft edges of every direction, a ramp, noise patches."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    img = 90 + 40 * np.sin(xx / 17.0) * np.cos(yy / 13.0)
    img[(xx - w / 3) ** 2 + (yy - h / 2) ** 2 < (min(h, w) / 5) ** 2] = 200
    img[h // 6:h // 3, w // 2:w - w // 8] = 30
    img[(xx + yy > (h + w) // 2) & (xx - yy < w // 3) & (yy > h // 2)] += 45
    img[2 * h // 3:, :w // 4] = (xx[2 * h // 3:, :w // 4] * 255.0 / max(w // 4, 1))
    n0, n1 = h // 10, h // 4
    img[n0:n1, w // 12:w // 3] += rng.integers(-35, 36, (n1 - n0, w // 3 - w // 12))
    img += rng.integers(-2, 3, (h, w))
    return np.clip(img, 0, 255).astype(np.uint8)


def canny_numpy(gray, low, high):
    """The same rules from their description, with vectorised numpy: returns (edges, unsure) where
    `unsure` marks pixels whose gradient direction lies within 1e-3 of a sector border (the integer
    tangent test of the restatement and the float test here may then differ)."""
    g = np.pad(gray.astype(np.int64), 1, mode="edge")
    h, w = gray.shape
    win = lambda dy, dx: g[1 + dy:1 + dy + h, 1 + dx:1 + dx + w]
    gx = (win(-1, 1) + 2 * win(0, 1) + win(1, 1)) - (win(-1, -1) + 2 * win(0, -1) + win(1, -1))
    gy = (win(1, -1) + 2 * win(1, 0) + win(1, 1)) - (win(-1, -1) + 2 * win(-1, 0) + win(-1, 1))
    m = np.abs(gx) + np.abs(gy)
    mp = np.pad(m, 1)
    nb = lambda dy, dx: mp[1 + dy:1 + dy + h, 1 + dx:1 + dx + w]
    ax, ay = np.abs(gx).astype(np.float64), np.abs(gy).astype(np.float64)
    t22, t67 = np.tan(np.pi / 8), np.tan(3 * np.pi / 8)
    horiz = ay < t22 * ax
    vert = ay > t67 * ax
    diag = ~horiz & ~vert
    same = (gx ^ gy) >= 0  # gradient along the main diagonal
    keep = np.zeros((h, w), bool)
    keep |= horiz & (m > nb(0, -1)) & (m >= nb(0, 1))
    keep |= vert & (m > nb(-1, 0)) & (m >= nb(1, 0))
    keep |= diag & same & (m > nb(-1, -1)) & (m > nb(1, 1))
    keep |= diag & ~same & (m > nb(-1, 1)) & (m > nb(1, -1))
    cand = keep & (m > low)
    strong = cand & (m > high)
    lab, n = ndimage.label(cand, structure=np.ones((3, 3)))
    good = np.zeros(n + 1, bool)
    good[np.unique(lab[strong])] = True
    good[0] = False
    edges = good[lab]
    unsure = (np.abs(ay - t22 * ax) <= 1e-3 * (ax + 1)) | (np.abs(ay - t67 * ax) <= 1e-3 * (ax + 1))
    return edges.astype(np.uint8) * 255, unsure


# ---------------------------------------------------------------- oracle, CPU

def test_canny_oracle_known_answers():
    flat = np.full((20, 30), 77, np.uint8)
    assert not orc.canny(flat, 50, 150).any()
    # a vertical step of 100: Sobel gives 400 on the two columns next to the step; the tie goes to
    # the left one (`>` to the left, `>=` to the right)
    step = np.zeros((12, 16), np.uint8)
    step[:, 8:] = 100
    e = orc.canny(step, 50, 150)
    assert (e[:, 7] == 255).all() and e.sum() == 255 * 12
    # a horizontal step: the upper row of the pair
    e = orc.canny(step.T.copy(), 50, 150)
    assert (e[7, :] == 255).all() and e.sum() == 255 * 12
    # thresholds are exclusive: magnitude 400 is no edge at high = 400, and weak-only pixels vanish
    assert not orc.canny(step, 50, 400).any()
    assert orc.canny(step, 399, 399.9).sum() == 255 * 12
    # swapped thresholds are put in order
    assert np.array_equal(orc.canny(step, 150, 50), orc.canny(step, 50, 150))


def test_canny_oracle_hysteresis_follows_weak_chains():
    # a bright line whose contrast fades along its length: strong at one end only
    img = np.full((40, 120), 60, np.uint8)
    contrast = np.linspace(80, 4, 120).astype(np.uint8)
    img[20, :] = 60 + contrast
    low, high = 40, 200
    e = orc.canny(img, low, high)
    ref, unsure = canny_numpy(img, low, high)
    assert np.array_equal(e, ref)
    cols = np.nonzero(e.any(axis=0))[0]
    # Sobel next to a one-pixel line of contrast c: |gy| = 4c; above `high` for c > 50 (columns below
    # 47), above `low` for c > 10 (columns up to about 109): the chain is kept as far as it stays
    # above `low`
    assert cols.min() == 0 and 100 < cols.max() < 115
    assert not orc.canny(img, low, 2000).any()


@pytest.mark.parametrize("h,w,seed,low,high", [(97, 131, 1, 50, 150), (64, 64, 2, 40, 120), (200, 310, 3, 50, 150),
                                               (33, 500, 4, 20, 60), (5, 7, 5, 10, 30), (1, 40, 6, 10, 30),
                                               (40, 1, 7, 10, 30), (120, 90, 8, 0, 0)])
def test_canny_oracle_against_numpy_statement(h, w, seed, low, high):
    img = picture(h, w, seed)
    e = orc.canny(img, low, high)
    ref, unsure = canny_numpy(img, low, high)
    if not unsure.any():
        assert np.array_equal(e, ref)
    else:
        # differences may only start at pixels whose direction is on a sector border
        diff = e != ref
        lab, n = ndimage.label(diff | unsure, structure=np.ones((3, 3)))
        touched = np.unique(lab[unsure])
        assert np.isin(lab[diff], touched).all()
    assert set(np.unique(e)) <= {0, 255}


# ---------------------------------------------------------------- HIP path, GPU

@pytest.fixture(scope="module")
def ctx():
    c = rtvm.Context()
    yield c
    c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("h,w,seed,low,high", [(97, 131, 1, 50, 150), (64, 64, 2, 40, 120), (16, 64, 3, 50, 150),
                                               (17, 65, 4, 30, 90), (5, 7, 5, 10, 30), (1, 40, 6, 10, 30),
                                               (40, 1, 7, 10, 30), (720, 1280, 8, 50, 150), (333, 517, 9, 40, 120),
                                               (200, 310, 10, 0, 0), (200, 310, 11, 99.5, 100.5)])
def test_hip_canny_matches_oracle(ctx, h, w, seed, low, high):
    img = picture(h, w, seed)
    assert np.array_equal(ctx.canny(img, low, high), orc.canny(img, low, high))


@pytest.mark.gpu
def test_hip_canny_weak_chain_across_many_tiles(ctx):
    # a weak step edge along a sine curve, strong over its first columns only: hysteresis has to carry
    # the mark along the whole curve, which crosses the border between two rows of tiles (y = 48) 24
    # times and eleven columns of tiles
    h, w = 100, 700
    yy, xx = np.mgrid[0:h, 0:w]
    curve = 48 + 7 * np.sin(xx / 9.0)
    img = np.where(yy > curve, 80, 50).astype(np.uint8)
    img[(yy > curve) & (xx < 12)] = 255
    low, high = 30, 500
    e = ctx.canny(img, low, high)
    assert np.array_equal(e, orc.canny(img, low, high))
    assert e[:, w - 1].any() and e.sum() // 255 > 800   # the mark reached the far end of the curve
    assert not ctx.canny(img, low, 2000).any()


@pytest.mark.gpu
def test_hip_canny_noise_and_arguments(ctx):
    rng = np.random.default_rng(3)
    noise = rng.integers(0, 256, (211, 307)).astype(np.uint8)
    for low, high in [(50, 150), (300, 600), (150, 50), (0, 2000)]:
        assert np.array_equal(ctx.canny(noise, low, high), orc.canny(noise, low, high))
    assert np.array_equal(rtvm.canny_edges(noise, 40, 120, ctx=ctx), orc.canny(noise, 40, 120))
    with pytest.raises(ValueError):
        ctx.canny(np.zeros((4, 4, 3), np.uint8), 1, 2)
    with pytest.raises(rtvm.RtvmError):
        ctx.canny(noise, float("nan"), 2)
