"""warp / distance transform / blur / blend: oracle vs numpy twins (CPU), HIP vs oracle (GPU)."""
import os
import sys

import numpy as np
import pytest

import oracle_ffi as orc
sys.path.insert(0, os.path.join(orc.ROOT, "oracle"))
import np_twin  # noqa: E402
import rtvm  # noqa: E402
from rtvm import synth  # noqa: E402

FH, FW, CH, CW = 96, 128, 192, 160


def frames_small(n=6, zeros=True):
    world = synth.make_world(400, 400, seed=5)
    frames = []
    Gs = synth.camera_path(n, 40.0, 260.0, seed=7, step=6.0, rot_deg=0.7, zoom=2e-3)
    for G in Gs:
        f = synth.render_frame(world, G, FH, FW)
        frames.append(f)
    if zeros:                      # true zero pixels: holes in mask_new
        frames[2][10:30, 20:50] = 0
        frames[3][:, :5] = 0
        frames[4][40:44, 60:64, 1] = 0   # one zero channel only: still "inside"
    return frames, Gs


def abs_h(Gs, row_off, col_off):
    """Ground-truth absolute homographies frame_t -> canvas."""
    T = np.array([[1.0, 0, col_off], [0, 1.0, row_off], [0, 0, 1.0]])
    out = []
    for G in Gs:
        H = T @ np.linalg.inv(Gs[0]) @ G
        out.append(H / H[2, 2])
    return out


HS = {
    "identity": np.eye(3),
    "shift": np.array([[1.0, 0, 17], [0, 1, 29], [0, 0, 1]]),
    "half": np.array([[1.0, 0, 10.5], [0, 1, 20.5], [0, 0, 1]]),
    "affine": np.array([[0.96, -0.12, 21.3], [0.1, 1.04, 33.7], [0, 0, 1]]),
    "persp": np.array([[1.02, 0.03, 12.2], [-0.02, 0.97, 40.1], [4e-4, -3e-4, 1]]),
    "clipped": np.array([[1.0, 0.02, -40.5], [0.0, 1.0, 150.25], [0, 0, 1]]),
    "offcanvas": np.array([[1.0, 0, 500.0], [0, 1.0, 20], [0, 0, 1]]),
    "singular": np.array([[1.0, 2.0, 3.0], [2.0, 4.0, 6.0], [0, 0, 1]]),
}


# ------------------------------------------------------------------ CPU
def test_dt_oracle_equals_closed_form():
    rng = np.random.default_rng(0)
    for trial in range(25):
        h, w = int(rng.integers(1, 48)), int(rng.integers(1, 48))
        p = rng.choice([0.0, 0.02, 0.3, 0.9, 1.0])
        m = (rng.random((h, w)) >= p).astype(np.uint8) * 255
        assert np.array_equal(orc.distance_transform(m), np_twin.chamfer_dt_closed_form(m))


def test_dt_hand_values():
    m = np.full((5, 7), 255, np.uint8)
    m[2, 1] = 0
    d = orc.distance_transform(m)
    a, b = 62587, 89738
    assert d[2, 1] == 0
    assert d[2, 4] == np.float32(3 * a) / np.float32(65536)
    assert d[0, 3] == np.float32(2 * b) / np.float32(65536)
    assert d[4, 6] == np.float32(2 * b + 3 * a) / np.float32(65536)


def test_blur_oracle_vs_scipy():
    from scipy.ndimage import correlate1d
    rng = np.random.default_rng(1)
    src = rng.random((40, 70)).astype(np.float32)
    k = np.exp(-0.5 * ((np.arange(31) - 15.0) / 5.0) ** 2)
    k = (k / k.sum()).astype(np.float32)
    ref = correlate1d(correlate1d(src.astype(np.float64), k.astype(np.float64), axis=1, mode="mirror"),
                      k.astype(np.float64), axis=0, mode="mirror")
    got = orc.gaussian_blur31(src)
    assert np.abs(got - ref).max() < 2e-6
    assert abs(float(orc.gaussian_blur31(np.ones((33, 35), np.float32)).mean()) - 1.0) < 1e-6


@pytest.mark.parametrize("fixed", [0, 1])
def test_warp_oracle_known_answers(fixed):
    rng = np.random.default_rng(2)
    f = rng.integers(1, 256, (20, 30, 3)).astype(np.uint8)
    o = orc.warp_perspective(f, np.eye(3), 40, 50, fixed)
    assert np.array_equal(o[:20, :30], f) and o[20:].max() == 0 and o[:, 30:].max() == 0
    H = np.eye(3)
    H[0, 2], H[1, 2] = 7, 11
    o = orc.warp_perspective(f, H, 40, 50, fixed)
    assert np.array_equal(o[11:31, 7:37], f) and o[:11].max() == 0 and o[:, :7].max() == 0
    H = np.eye(3)
    H[0, 2] = 0.5
    o = orc.warp_perspective(f, H, 40, 50, fixed)
    exp = (f[:, :-1].astype(np.float64) + f[:, 1:]) / 2
    assert np.abs(o[:20, 1:30] - exp).max() <= 0.5
    # the two arithmetic modes agree to one grey level on a smooth affine map
    a = orc.warp_perspective(f, HS["affine"], 64, 64, 0).astype(int)
    b = orc.warp_perspective(f, HS["affine"], 64, 64, 1).astype(int)
    assert np.abs(a - b).max() <= 8 and np.mean(a != b) < 0.6


@pytest.mark.parametrize("fixed", [0, 1])
def test_blend_oracle_equals_numpy_statements(fixed):
    """oracle warp_blend == main.py:878-927 executed in numpy on a float64 canvas."""
    frames, Gs = frames_small()
    row_off, col_off = CH - FH, int(CW / 2 - FW / 2)
    Hs = abs_h(Gs, row_off, col_off)
    canvas_u8 = np.zeros((CH, CW, 3), np.uint8)
    canvas_u8[row_off:row_off + FH, col_off:col_off + FW] = frames[0]
    canvas_f64 = canvas_u8.astype(np.float64)
    for f, H in zip(frames[1:], Hs[1:]):
        warped = orc.warp_perspective(f, H, CH, CW, fixed)
        canvas_f64 = np_twin.blend_numpy(canvas_f64, warped, orc.distance_transform, orc.gaussian_blur31)
        canvas_u8 = orc.warp_blend(canvas_u8, f, H, fixed)
        assert canvas_f64.dtype == np.float64
        assert np.array_equal(canvas_f64, canvas_u8.astype(np.float64))
    assert (canvas_u8.max(axis=2) > 0).mean() > 0.5


def test_blend_no_overlap_and_offcanvas():
    frames, _ = frames_small(2, zeros=False)
    canvas = np.zeros((CH, CW, 3), np.uint8)
    H = np.array([[0.4, 0, 5.0], [0, 0.4, 6.0], [0, 0, 1]])
    c1 = orc.warp_blend(canvas, frames[0], H, 1)
    assert np.array_equal(c1, orc.warp_perspective(frames[0], H, CH, CW, 1))
    c2 = orc.warp_blend(c1, frames[1], HS["offcanvas"], 1)
    assert np.array_equal(c1, c2)


# ------------------------------------------------------------------ GPU
@pytest.fixture(scope="module")
def ctxs():
    out = {0: rtvm.Context(FH, FW, CH, CW, flags=0), 1: rtvm.Context(FH, FW, CH, CW, flags=rtvm._ffi.FLAG_WARP_FIXED)}
    yield out
    for c in out.values():
        c.close()


@pytest.mark.gpu
def test_hip_dt_bit_exact(ctxs):
    rng = np.random.default_rng(3)
    c = ctxs[0]
    shapes = [(1, 1), (1, 37), (29, 1), (17, 23), (64, 64), (100, 150), (CH, CW), (150, 200)]
    for (h, w) in shapes:
        for p in (0.0, 0.01, 0.3, 0.95, 1.0):
            m = (rng.random((h, w)) >= p).astype(np.uint8) * 255
            assert np.array_equal(c.distance_transform(m), orc.distance_transform(m)), (h, w, p)
    # blob-shaped masks like the mosaic footprints
    m = np.zeros((CH, CW), np.uint8)
    m[30:150, 20:140] = 255
    m[60:70, 50:60] = 0
    assert np.array_equal(c.distance_transform(m), orc.distance_transform(m))


@pytest.mark.gpu
def test_hip_dt_wide_and_tall_images():
    """Rows wider than one workgroup pass of the row scan (2048 pixels), pitches that are not
    multiples of 8, more than 16 and more than 32 chunks of 32 rows, sparse zeros far apart."""
    rng = np.random.default_rng(5)
    c = rtvm.Context(48, 64, 700, 5200)
    try:
        for (h, w) in [(40, 5000), (33, 4099), (3, 2049), (700, 61), (600, 40), (520, 515)]:
            for p in (0.0005, 0.02, 0.7):
                m = (rng.random((h, w)) >= p).astype(np.uint8) * 255
                assert np.array_equal(c.distance_transform(m), orc.distance_transform(m)), (h, w, p)
            m = np.full((h, w), 255, np.uint8)
            m[h - 1, 0] = 0   # a single zero in a corner: every chunk and pass carries it across
            assert np.array_equal(c.distance_transform(m), orc.distance_transform(m)), (h, w, "corner")
            m[:] = 255        # no zero at all
            assert np.array_equal(c.distance_transform(m), orc.distance_transform(m)), (h, w, "none")
    finally:
        c.close()


@pytest.mark.gpu
def test_hip_dt_tile_borders():
    """Image sizes around the scan kernels' tile (32 rows x 64 columns): a single zero in every corner
    and in the middle of a tile border, so that every distance arrives through the summaries of other
    tiles, along rows, along columns and along both."""
    rng = np.random.default_rng(9)
    c = rtvm.Context(48, 64, 140, 280)
    try:
        for h in (31, 32, 33, 64, 65, 97):
            for w in (63, 64, 65, 127, 128, 129, 200):
                spots = [(0, 0), (0, w - 1), (h - 1, 0), (h - 1, w - 1), (min(31, h - 1), min(63, w - 1)),
                         (min(32, h - 1), min(64, w - 1)), (h // 2, w // 2)]
                for (y, x) in spots:
                    m = np.full((h, w), 255, np.uint8)
                    m[y, x] = 0
                    assert np.array_equal(c.distance_transform(m), orc.distance_transform(m)), (h, w, y, x)
                m = (rng.random((h, w)) >= 0.004).astype(np.uint8) * 255
                assert np.array_equal(c.distance_transform(m), orc.distance_transform(m)), (h, w, "sparse")
    finally:
        c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("fixed", [0, 1])
@pytest.mark.parametrize("name", sorted(HS))
def test_hip_warp_perspective_exact(ctxs, fixed, name):
    frames, _ = frames_small()
    for f in (frames[0], frames[2]):
        got = ctxs[fixed].warp_perspective(f, HS[name])
        ref = orc.warp_perspective(f, HS[name], CH, CW, fixed)
        assert np.array_equal(got, ref)


@pytest.mark.gpu
@pytest.mark.parametrize("fixed", [0, 1])
def test_hip_warp_blend_sequence_exact(ctxs, fixed):
    frames, Gs = frames_small()
    row_off, col_off = CH - FH, int(CW / 2 - FW / 2)
    Hs = abs_h(Gs, row_off, col_off)
    c = ctxs[fixed]
    c.canvas_init(frames[0], row_off, col_off)
    ref = np.zeros((CH, CW, 3), np.uint8)
    ref[row_off:row_off + FH, col_off:col_off + FW] = frames[0]
    assert np.array_equal(c.canvas_read(), ref)
    seq = list(zip(frames[1:], Hs[1:])) + [(frames[1], HS["clipped"]), (frames[2], HS["offcanvas"]),
                                           (frames[3], HS["persp"])]
    for i, (f, H) in enumerate(seq):
        c.warp_blend(f, H)
        ref = orc.warp_blend(ref, f, H, fixed)
        got = c.canvas_read()
        assert np.array_equal(got, ref), "frame %d: %d pixels differ" % (i, int((got != ref).any(axis=2).sum()))


@pytest.mark.gpu
@pytest.mark.parametrize("fixed", [0, 1])
def test_hip_warp_blend_region_much_larger_than_the_frame(fixed):
    """The launch grids of the warp / blur kernels are sized for a region of about the frame's size;
    a frame magnified 3.5 times covers many more tiles (taken by the kernels' loops and overflow
    launches)."""
    fh, fw, ch, cw = 48, 64, 400, 1100
    rng = np.random.default_rng(11)
    frames = [rng.integers(1, 256, (fh, fw, 3), dtype=np.uint8) for _ in range(3)]
    c = rtvm.Context(fh, fw, ch, cw, flags=rtvm._ffi.FLAG_WARP_FIXED if fixed else 0)
    try:
        c.canvas_init(frames[0], 100, 300)
        ref = np.zeros((ch, cw, 3), np.uint8)
        ref[100:100 + fh, 300:300 + fw] = frames[0]
        for f, H in [(frames[1], np.array([[3.5, 0.1, 250.0], [-0.05, 3.4, 60.0], [0, 0, 1]])),
                     (frames[2], np.array([[16.0, 0.0, 20.0], [0.0, 7.5, 15.0], [0, 0, 1]])),
                     (frames[1], np.array([[1.0, 0.0, 310.0], [0.0, 1.0, 110.0], [0, 0, 1]]))]:
            c.warp_blend(f, H)
            ref = orc.warp_blend(ref, f, H, fixed)
            got = c.canvas_read()
            assert np.array_equal(got, ref), "%d pixels differ" % int((got != ref).any(axis=2).sum())
    finally:
        c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("cw", [161, 162, 163])
def test_hip_warp_blend_canvas_width_not_a_multiple_of_four(cw):
    """Canvas rows that do not start on 16-byte boundaries: the commit kernel with one pixel per lane and every
    plane at an odd pitch (a multiple of 4 takes the kernel with four pixels per lane)."""
    fh, fw, ch = 48, 64, 150
    rng = np.random.default_rng(cw)
    frames = [rng.integers(0, 256, (fh, fw, 3), dtype=np.uint8) for _ in range(4)]
    frames[2][10:20, 5:30] = 0   # zero pixels inside a frame: holes in its mask
    c = rtvm.Context(fh, fw, ch, cw)
    try:
        c.canvas_init(frames[0], 40, 50)
        ref = np.zeros((ch, cw, 3), np.uint8)
        ref[40:40 + fh, 50:50 + fw] = frames[0]
        for f, H in [(frames[1], np.array([[1.02, 0.01, 58.0], [-0.01, 0.99, 44.0], [0, 0, 1]])),
                     (frames[2], np.array([[0.98, -0.02, 66.0], [0.02, 1.01, 37.0], [1e-5, 0, 1]])),
                     (frames[3], np.array([[1.0, 0.0, float(cw - 40)], [0.0, 1.0, 80.0], [0, 0, 1]]))]:   # clipped at the right edge
            c.warp_blend(f, H)
            ref = orc.warp_blend(ref, f, H, 0)
            got = c.canvas_read()
            assert np.array_equal(got, ref), "%d pixels differ" % int((got != ref).any(axis=2).sum())
    finally:
        c.close()


@pytest.mark.gpu
def test_hip_warp_blend_no_overlap(ctxs):
    frames, _ = frames_small(2, zeros=False)
    c = ctxs[1]
    c.canvas_init(np.zeros_like(frames[0]), 0, 0)
    H = np.array([[0.4, 0, 5.0], [0, 0.4, 6.0], [0, 0, 1]])
    c.warp_blend(frames[0], H)
    assert np.array_equal(c.canvas_read(), orc.warp_perspective(frames[0], H, CH, CW, 1))


@pytest.mark.gpu
def test_hip_warp_blend_requires_init():
    c = rtvm.Context(FH, FW, CH, CW)
    with pytest.raises(rtvm.RtvmError):
        c.warp_blend(np.zeros((FH, FW, 3), np.uint8), np.eye(3))
    c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("ch,cw,fh,fw", [(61, 53, 24, 30), (192, 161, 96, 128), (1440, 1536, 720, 1280)])
def test_hip_canvas_as_float64_is_the_u8_canvas_widened(ch, cw, fh, fw):
    """rtvm_canvas_read_f64 (the reference's output_img is float64 holding u8 values): every element, for canvases
    whose byte count is and is not a multiple of the vector width, above and below the size that goes onto threads."""
    rng = np.random.default_rng(ch)
    ctx = rtvm.Context(fh, fw, ch, cw, nfeatures=300)
    first = rng.integers(0, 256, (fh, fw, 3)).astype(np.uint8)
    ctx.canvas_init(first, ch - fh, (cw - fw) // 2)
    u8 = ctx.canvas_read()
    f64 = ctx.canvas_read_f64()
    again = ctx.canvas_read_f64()
    ctx.close()
    assert f64.dtype == np.float64 and f64.shape == (ch, cw, 3)
    assert np.array_equal(f64, u8.astype(np.float64)) and np.array_equal(again, f64)
    assert u8[ch - fh:, (cw - fw) // 2:(cw - fw) // 2 + fw].tobytes() == first.tobytes()


@pytest.mark.gpu
@pytest.mark.parametrize("fused", ["0", "1"])
@pytest.mark.parametrize("cw,ch", [(160, 192), (163, 171), (257, 100)])
def test_hip_blend_column_pass_inside_the_commit_or_as_a_kernel_of_its_own(fused, cw, ch, monkeypatch):
    """The two ways the canvas half ends -- k_blur_cols + k_commit(4), or k_commit_cols with the column sums inside the
    commit (the default for frames of 0.8 megapixels and more) -- against the oracle: a sequence of overlapping frames, canvas
    sizes that are and are not multiples of the tiles (64 x 48 and 64 x 8) and of four, frames at the canvas' top and bottom rows
    (reflected blur rows)."""
    monkeypatch.setenv("RTVM_FUSED_COMMIT", fused)
    frames, _ = frames_small(6)
    ctx = rtvm.Context(FH, FW, ch, cw, nfeatures=300)
    row_off, col_off = max(0, ch - FH), max(0, (cw - FW) // 2)
    ctx.canvas_init(frames[0], row_off, col_off)
    canvas = np.zeros((ch, cw, 3), np.uint8)
    canvas[row_off:row_off + FH, col_off:col_off + FW] = frames[0][:ch - row_off, :cw - col_off]
    Hs = [np.array([[1.0, 0.0, col_off + 3.5 * t], [0.0, 1.0, row_off - 9.25 * t], [0.0, 0.0, 1.0]]) for t in range(1, 6)]
    Hs[3] = np.array([[0.98, 0.03, col_off - 6.0], [-0.03, 0.98, -5.0], [1e-5, -2e-5, 1.0]])     # up to the top rows
    Hs[4] = np.array([[1.02, 0.0, col_off + 1.0], [0.0, 1.02, ch - FH * 0.6], [0.0, 0.0, 1.0]])   # over the bottom rows
    for f, H in zip(frames[1:], Hs):
        ctx.warp_blend(f, H)
        canvas = orc.warp_blend(canvas, f, H, 0)
    got = ctx.canvas_read()
    ctx.close()
    assert np.array_equal(got, canvas)
    assert got[:8].any() and got[-8:].any()
