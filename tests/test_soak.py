"""A long video through the three ways of running the path: every ring of the library (ingest slots, report
slots, tickets, slot sets, the sequence counters behind them) wraps many times, and the three must still
agree on every status, the final homography and every canvas pixel."""
import numpy as np
import pytest

import rtvm
from rtvm import synth


@pytest.mark.gpu
@pytest.mark.parametrize("detector,nf,n,fh,fw", [("orb", 500, 1300, 240, 320), ("sift", 300, 400, 240, 320),
                                                 ("orb", 1000, 1024, 720, 1280)])
def test_long_video_pipe_process_video_and_frame_by_frame_agree(detector, nf, n, fh, fw, capsys):
    import torch
    ch, cw = 2 * fh, int(1.2 * fw)
    Gs, wh, ww = synth.bounded_path(n + 1, fh, fw, seed=11, margin=100)
    world = synth.make_world(wh, ww, seed=4)
    frames = synth.render_video_torch(world, Gs, fh, fw, torch.device("cuda", 0))
    torch.cuda.synchronize()
    host = frames.cpu().numpy()

    # frame by frame, every call synchronous
    sync = rtvm.VideMosaic(host[0], detector_type=detector, nfeatures=nf, pipeline_batch=0)
    assert sync.output_img.shape == (ch, cw, 3)
    for t in range(1, n + 1):
        sync.process_frame(host[t], t)
    out_sync = capsys.readouterr().out
    want_H, want_canvas = sync.H.copy(), sync.output_img_u8()
    want_hist = [h.copy() for h in sync.homography_history]
    sync.close()
    assert want_canvas[:fh // 2].any()          # the camera climbed over the canvas

    # the pipeline behind process_frame, host frames through the ingest ring
    vm = rtvm.VideMosaic(host[0], detector_type=detector, nfeatures=nf)
    assert vm._deferred
    for t in range(1, n + 1):
        vm.process_frame(host[t], t)
    assert np.array_equal(vm.H, want_H)
    assert capsys.readouterr().out == out_sync
    assert np.array_equal(vm.output_img_u8(), want_canvas)
    assert len(vm.homography_history) == len(want_hist)
    assert all(np.array_equal(a, b) for a, b in zip(vm.homography_history, want_hist))
    stats = vm._pipe.stats
    assert stats["frames"] == n
    vm.close()

    # the whole loop inside the library, frames resident on the device
    det = rtvm._ffi.DET_SIFT if detector == "sift" else rtvm._ffi.DET_ORB
    ctx = rtvm.Context(fh, fw, ch, cw, nfeatures=nf, detector=det)
    stride, base = fh * fw * 3, frames.data_ptr()
    ctx.stream_begin(base, ch - fh, int(cw / 2 - fw / 2))
    status = ctx.process_video([base + (1 + j) * stride for j in range(n)], batch=32)
    _, _, _, H = ctx.frame_status()
    canvas = ctx.canvas_read()
    ctx.close()
    assert len(status) == n
    assert int((status != 0).sum()) == out_sync.count("пропуск кадра") + out_sync.count("Невалидная гомография")
    assert np.array_equal(H, want_H)
    assert np.array_equal(canvas, want_canvas)
