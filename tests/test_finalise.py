"""Mosaic finalisation (reference main.py:980-1038): crop_black_areas' rectangle and scale_to_screen's
cv2.resize(INTER_LINEAR).  CPU tests pin the oracle against numpy statements; GPU tests compare the HIP
path with the oracle bit for bit, through the C ABI and through the drop-in functions."""
import importlib

import numpy as np
import pytest

import oracle_ffi as orc

rtvm = importlib.import_module("real-time-video-mosaic_amd")


def mosaic_like(h, w, seed, box):
    """Dark canvas (values that stay at or below gray 15) with a bright textured rectangle."""
    rng = np.random.default_rng(seed)
    img = rng.integers(0, 12, (h, w, 3)).astype(np.uint8)
    y0, y1, x0, x1 = box
    img[y0:y1, x0:x1] = rng.integers(40, 256, (y1 - y0, x1 - x0, 3))
    return img


def test_crop_rect_oracle_is_the_bounding_box_of_bright_pixels():
    img = mosaic_like(90, 130, 1, (17, 63, 29, 101))
    assert orc.crop_rect(img, 15) == (29, 17, 72, 46)
    # numpy statement of main.py:991-997
    gray = orc.gray(img)
    ys, xs = np.nonzero(gray > 15)
    assert orc.crop_rect(img, 15) == (xs.min(), ys.min(), xs.max() - xs.min() + 1, ys.max() - ys.min() + 1)
    assert orc.crop_rect(np.zeros((20, 30, 3), np.uint8), 15) is None
    one = np.zeros((20, 30, 3), np.uint8)
    one[7, 11] = 200
    assert orc.crop_rect(one, 15) == (11, 7, 1, 1)


def test_resize_oracle_known_answers():
    rng = np.random.default_rng(2)
    img = rng.integers(0, 256, (23, 31, 3)).astype(np.uint8)
    assert np.array_equal(orc.resize_linear(img, 31, 23), img)  # same size: every weight is 1
    flat = np.full((9, 14, 3), 77, np.uint8)
    assert (orc.resize_linear(flat, 40, 33) == 77).all() and (orc.resize_linear(flat, 5, 4) == 77).all()
    # against float bilinear with the same sample positions: the 11-bit coefficients cost at most one level
    for nw, nh in [(62, 46), (47, 20), (13, 9), (100, 7)]:
        got = orc.resize_linear(img, nw, nh).astype(np.float64)
        fx = np.clip((np.arange(nw) + 0.5) * 31 / nw - 0.5, 0, 30)
        fy = np.clip((np.arange(nh) + 0.5) * 23 / nh - 0.5, 0, 22)
        x0, y0 = np.minimum(np.floor(fx).astype(int), 29), np.minimum(np.floor(fy).astype(int), 21)
        ax, ay = (fx - x0)[None, :, None], (fy - y0)[:, None, None]
        f = img.astype(np.float64)
        ref = ((f[y0][:, x0] * (1 - ax) + f[y0][:, x0 + 1] * ax) * (1 - ay) +
               (f[y0 + 1][:, x0] * (1 - ax) + f[y0 + 1][:, x0 + 1] * ax) * ay)
        assert np.abs(got - ref).max() <= 1.0


@pytest.mark.gpu
@pytest.mark.parametrize("shape,box", [((90, 130), (17, 63, 29, 101)), ((64, 64), (0, 64, 0, 64)),
                                       ((257, 191), (200, 257, 5, 70)), ((40, 300), (39, 40, 299, 300))])
def test_hip_crop_rect_matches_oracle(shape, box):
    img = mosaic_like(shape[0], shape[1], 3, box)
    ctx = rtvm.Context()
    assert ctx.crop_rect(img, 15) == orc.crop_rect(img, 15)
    assert ctx.crop_rect(img, 250) == orc.crop_rect(img, 250)
    assert ctx.crop_rect(np.zeros((shape[0], shape[1], 3), np.uint8), 15) is None
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("size", [(62, 46), (47, 20), (13, 9), (100, 7), (31, 23), (1, 1), (640, 360)])
def test_hip_resize_matches_oracle(size):
    rng = np.random.default_rng(4)
    img = rng.integers(0, 256, (23, 31, 3)).astype(np.uint8)
    ctx = rtvm.Context()
    assert np.array_equal(ctx.resize_linear(img, size[0], size[1]), orc.resize_linear(img, size[0], size[1]))
    big = rng.integers(0, 256, (480, 854, 3)).astype(np.uint8)
    assert np.array_equal(ctx.resize_linear(big, 1920, 1079), orc.resize_linear(big, 1920, 1079))
    ctx.close()


@pytest.mark.gpu
def test_hip_finalisation_of_a_device_canvas_and_the_drop_in_functions():
    first = np.zeros((96, 128, 3), np.uint8)
    first[:] = np.random.default_rng(5).integers(30, 256, (96, 128, 3))
    vm = rtvm.VideMosaic(first, detector_type="orb", visualize=False, show_intermediate=False)
    canvas = vm.output_img_u8()
    rect = vm._ctx.canvas_crop_rect(15)
    assert rect == orc.crop_rect(canvas, 15) and rect is not None
    x, y, w, h = rect
    got = vm._ctx.canvas_read_scaled(x, y, w, h, 300, 200)
    assert np.array_equal(got, orc.resize_linear(canvas[y:y + h, x:x + w], 300, 200))
    assert np.array_equal(vm.preview(), orc.resize_linear(canvas, 400, 300))   # what gui.py:143-155 displays
    assert np.array_equal(vm.preview(64, 50), orc.resize_linear(canvas, 64, 50))
    # the reference's two functions (main.py:980-1038), on a numpy image and on the mosaic object
    cropped = rtvm.crop_black_areas(canvas)
    assert np.array_equal(cropped, canvas[y + 5:y + 5 + h - 10, x + 5:x + 5 + w - 10])
    assert np.array_equal(rtvm.crop_black_areas(vm), cropped)
    assert np.array_equal(rtvm.crop_black_areas(vm.output_img), cropped)  # float64 input, as main.py:1649 passes
    black = np.zeros((10, 12, 3), np.uint8)
    assert rtvm.crop_black_areas(black) is black
    scaled = rtvm.scale_to_screen(cropped)
    s = min(1920 / cropped.shape[1], 1080 / cropped.shape[0])
    nw, nh = max(1, int(cropped.shape[1] * s)), max(1, int(cropped.shape[0] * s))
    assert scaled.shape == (nh, nw, 3) and np.array_equal(scaled, orc.resize_linear(cropped, nw, nh))
    assert rtvm.scale_to_screen(cropped, 64, 64).shape[:2] == (min(64, int(cropped.shape[0] * 64 / cropped.shape[1])), 64)
    vm.close()
