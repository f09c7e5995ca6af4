"""The configurations the benchmark quotes, at their own sizes, against the CPU restatement:
the 720p ORB-1000 video of bench.py through the batched device pipeline across episodes of
mispredicted canvas masks (retried canvas halves), 1080p with the reference's default detector
(SIFT, 2000 features) through the drop-in class, and the 4K frame on the 4096 x 16384 canvas
through the warp + blend stage."""
import numpy as np
import pytest

import oracle_ffi as orc
import rtvm
from rtvm import synth
from oracle_mosaic import OracleMosaic


def bench_video(n, fh, fw, device, path_seed=2, world_seed=0):
    """Frames 0..n of bench.py's video (same world, same camera path), on the device and on the host."""
    Gs, wh, ww = synth.bounded_path(n + 1, fh, fw, seed=path_seed)
    world = synth.make_world(wh, ww, seed=world_seed)
    frames = synth.render_video_torch(world, Gs, fh, fw, device)
    return frames, frames.cpu().numpy()


@pytest.mark.gpu
def test_bench_video_720p_batched_with_mask_retries_matches_oracle():
    """304 frames of the headline workload through Context.process_video(batch=32), as bench.py runs it.
    The camera sways back over the mosaic's rim around frames 64-95 and, after the turn at the top of its climb, around frames 256-290: blends clear canvas pixels there,
    the predicted masks of the frames behind are wrong, and their canvas halves are queued again.
    Statuses, the final homography and every canvas pixel equal the frame-by-frame oracle."""
    import torch
    fh, fw, ch, cw, nf, n = 720, 1280, 1440, 1536, 1000, 304
    frames, host = bench_video(n, fh, fw, torch.device("cuda", 0))
    torch.cuda.synchronize()
    ctx = rtvm.Context(fh, fw, ch, cw, nfeatures=nf)
    stride, base = fh * fw * 3, frames.data_ptr()
    ctx.stream_begin(base, ch - fh, int(cw / 2 - fw / 2))
    status = ctx.process_video([base + (1 + j) * stride for j in range(n)], batch=32)
    stats = dict(ctx.video_stats)
    assert stats["batches_with_pending_frames"] > 0 and stats["retries"] > 0 and stats["frames_retried"] > 0, stats
    _, _, nm, H = ctx.frame_status()
    canvas = ctx.canvas_read()
    ctx.close()
    om = OracleMosaic(host[0], nfeatures=nf, canvas_size=(ch, cw), glue="explicit")
    want = []
    for f in host[1:]:
        om.process_frame(f)
        want.append(om.status)
    assert status.tolist() == want == [0] * n
    assert nm == om.n_matches and nm > 300
    assert np.array_equal(H, om.H)
    assert np.array_equal(canvas, om.canvas)


@pytest.mark.gpu
def test_1080p_default_sift_2000_dropin_matches_oracle():
    """VideMosaic(first, nfeatures=2000) with the reference's default detector at 1080p: SIFT, knn-2 over
    2000 x 2000 x 128 + ratio test, RANSAC, warp + blend into the 2160 x 2304 canvas, frame by frame."""
    import torch
    fh, fw, nf, n = 1080, 1920, 2000, 3
    _, host = bench_video(n, fh, fw, torch.device("cuda", 0), path_seed=23, world_seed=5)
    vm = rtvm.VideMosaic(host[0], visualize=False, show_intermediate=False, nfeatures=nf)
    assert vm.detector_type == "sift" and vm.output_img.shape == (2160, 2304, 3)
    om = OracleMosaic(host[0], nfeatures=nf, detector="sift", glue="numpy")
    assert len(vm.kp_prev) == len(om.feat_prev["xy"]) >= nf
    for t in range(1, n + 1):
        vm.process_frame(host[t], t)
        om.process_frame(host[t])
        assert om.status == 0 and om.n_matches > 500
        assert [(m.queryIdx, m.trainIdx) for m in vm.matches] == list(zip(om.matches[0].tolist(), om.matches[1].tolist()))
        assert np.array_equal(vm.des_cur, om.feat_cur["desc"])
        assert np.array_equal(vm.H, om.H)
    assert np.array_equal(vm.output_img_u8(), om.canvas)
    vm.close()


@pytest.mark.gpu
def test_4k_frames_on_the_16k_canvas_warp_blend_matches_oracle():
    """rtvm_warp_blend with 2160 x 3840 frames on the 4096 x 16384 canvas: two overlapping frames (a shift with
    a little rotation and zoom, then a clipped one at the canvas edge) against the oracle's warp + blend."""
    import torch
    fh, fw, ch, cw = 2160, 3840, 4096, 16384
    _, host = bench_video(2, fh, fw, torch.device("cuda", 0), path_seed=31, world_seed=7)
    ctx = rtvm.Context(fh, fw, ch, cw, nfeatures=1000)
    row_off, col_off = ch - fh, int(cw / 2 - fw / 2)
    ctx.canvas_init(host[0], row_off, col_off)
    canvas = np.zeros((ch, cw, 3), np.uint8)
    canvas[row_off:row_off + fh, col_off:col_off + fw] = host[0]
    th = np.deg2rad(0.4)
    H1 = np.array([[1.003 * np.cos(th), -1.003 * np.sin(th), col_off + 310.25],
                   [1.003 * np.sin(th), 1.003 * np.cos(th), row_off - 420.5], [2e-6, -1e-6, 1.0]])
    H2 = np.array([[1.0, 0.0, cw - 2000.75], [0.0, 1.0, row_off - 1100.0], [0.0, 0.0, 1.0]])
    for f, H in ((host[1], H1), (host[2], H2)):
        ctx.warp_blend(f, H)
        canvas = orc.warp_blend(canvas, f, H, 0)
    got = ctx.canvas_read()
    ctx.close()
    assert np.array_equal(got, canvas)
    assert got[:, cw - 8:].any() and got[row_off - 400, col_off + 2000].any()
