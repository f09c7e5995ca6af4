"""End to end: the VideMosaic drop-in and the device-resident stream path against the
CPU restatement of process_frame, on a seeded synthetic panned video."""
import ctypes

import numpy as np
import pytest

import oracle_ffi as orc  # noqa: F401
import rtvm
from rtvm import synth
from oracle_mosaic import OracleMosaic

FH, FW = 240, 320


def video(n=8, seed=0):
    frames, Gs = synth.make_video(FH, FW, n, world_size=(520, 620), seed=seed)
    return frames, Gs


# ------------------------------------------------------------------ CPU
def test_glue_variants_agree():
    """np.linalg.det / np.matmul and the spelled-out forms agree: the decisions always, the product to the bit on
    a host whose BLAS sums a 3x3 product with fused multiply-adds (any x86-64 with FMA3 under numpy's OpenBLAS)."""
    rng = np.random.default_rng(0)
    f0 = np.zeros((80, 80, 3), np.uint8)
    a = OracleMosaic.__new__(OracleMosaic)
    b = OracleMosaic.__new__(OracleMosaic)
    a.glue, b.glue = "numpy", "explicit"
    for _ in range(50):
        H = np.eye(3) + rng.normal(0, 0.05, (3, 3))
        H[2, :2] = rng.normal(0, 5e-4, 2)
        H[:2, 2] = rng.normal(0, 30, 2)
        assert a.validate(H) == b.validate(H)
        assert np.allclose(a.chain(H, H.T), b.chain(H, H.T), rtol=1e-15, atol=1e-15)
        assert np.array_equal(a.chain(H, H.T), b.chain(H, H.T)), "np.matmul on this host does not fuse k = 0, 1, 2"


def test_glue_variants_agree_over_a_long_chain():
    """Three thousand frames' worth of validate / smooth / chain on random relative homographies, the reference's
    numpy statements against the spelled-out forms the device runs: every decision and every bit of every
    absolute homography."""
    rng = np.random.default_rng(5)
    a = OracleMosaic.__new__(OracleMosaic)
    b = OracleMosaic.__new__(OracleMosaic)
    for o, g in ((a, "numpy"), (b, "explicit")):
        o.glue, o.history, o.H_old = g, [], np.array([[1.0, 0.0, 32.0], [0.0, 1.0, 240.0], [0.0, 0.0, 1.0]])
    rejected = 0
    for t in range(3000):
        th = rng.normal(0, 0.004)
        s = 1.0 + rng.normal(0, 0.002)
        H = np.array([[s * np.cos(th), -s * np.sin(th), rng.normal(0, 3.0)],
                      [s * np.sin(th), s * np.cos(th), rng.normal(0, 3.0) - 0.002 * np.sin(t / 50.0)],
                      [rng.normal(0, 2e-6), rng.normal(0, 2e-6), 1.0]])
        if t % 97 == 0:
            H[0, 2] = 80.0       # a shaky frame
        ok = a.validate(H)
        assert ok == b.validate(H)
        rejected += not ok
        for o in (a, b):
            o.H = o.chain(o.H_old, o.smooth(H if ok else np.eye(3)))
            o.H_old = o.H
        assert np.array_equal(a.H, b.H), t
    assert rejected >= 30


def test_oracle_mosaic_tracks_ground_truth():
    frames, Gs = video(6)
    om = OracleMosaic(frames[0], nfeatures=500)
    T = np.eye(3)
    T[0, 2], T[1, 2] = om.h_offset, om.w_offset
    for t in range(1, 6):
        om.process_frame(frames[t])
        assert om.status == 0 and om.n_matches > 100
    H_gt = T @ np.linalg.inv(Gs[0]) @ Gs[5]
    H_gt /= H_gt[2, 2]
    corners = np.array([[0, 0, 1], [FW, 0, 1], [FW, FH, 1], [0, FH, 1]], float).T
    pe = om.H @ corners
    pg = H_gt @ corners
    err = np.abs(pe[:2] / pe[2] - pg[:2] / pg[2]).max()
    assert err < 4.0, err            # smoothing lags the motion by design (main.py:826-832)
    assert (om.canvas.max(axis=2) > 0).mean() > 0.42


# ------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize("warp_mode", ["float", "fixed"])
def test_videmosaic_dropin_matches_oracle(warp_mode, capsys):
    frames, _ = video(8)
    vm = rtvm.VideMosaic(frames[0], detector_type="orb", visualize=False, show_intermediate=False,
                         nfeatures=500, warp_mode=warp_mode)
    om = OracleMosaic(frames[0], nfeatures=500, fixed=(warp_mode == "fixed"), glue="numpy")
    assert vm.w_offset == om.w_offset and vm.h_offset == om.h_offset
    assert vm.output_img.dtype == np.float64 and vm.output_img.shape == (2 * FH, int(1.2 * FW), 3)
    assert np.array_equal(vm.output_img, om.canvas.astype(np.float64))
    for t in range(1, 8):
        assert vm.process_frame(frames[t], t) is None
        om.process_frame(frames[t])
        assert np.array_equal(vm.H, om.H)
        assert np.array_equal(vm.H_old, om.H_old)
        assert len(vm.matches) == om.n_matches
        assert np.array_equal(vm.output_img, om.canvas.astype(np.float64)), "frame %d" % t
    assert [m.queryIdx for m in vm.matches] == om.matches[0].tolist()
    assert np.array_equal(vm.des_prev, om.feat_prev["desc"])
    assert len(vm.kp_cur) == len(om.feat_cur["xy"]) and vm.kp_cur[3].pt == tuple(om.feat_cur["xy"][3].tolist())
    assert len(vm.homography_history) == 5
    vm.close()


@pytest.mark.gpu
@pytest.mark.parametrize("pipeline_batch", [0, 32])
def test_videmosaic_skip_rules(capsys, pipeline_batch):
    """With the reference's default display flags; frame-synchronous (pipeline_batch=0) and through the pipeline
    (flush() settles the frames taken so far: the warnings are printed then at the latest)."""
    frames, _ = video(3)
    vm = rtvm.VideMosaic(frames[0], detector_type="orb", nfeatures=300, pipeline_batch=pipeline_batch)
    assert vm.visualize and vm.show_intermediate and vm._deferred == (pipeline_batch > 0)
    before = vm.output_img.copy()
    H_old = vm.H_old.copy()
    des_prev = vm.des_prev.copy()
    flat = np.full_like(frames[0], 90)                     # featureless frame: < 4 matches
    vm.process_frame(flat, 1)
    vm.flush()
    assert "Недостаточно совпадений" in capsys.readouterr().out
    assert np.array_equal(vm.output_img, before) and np.array_equal(vm.H_old, H_old)
    assert np.array_equal(vm.des_prev, des_prev)           # state not advanced
    vm.process_frame(frames[1], 2)                         # the stream continues afterwards
    assert vm.H is not None and not np.array_equal(vm.output_img, before)
    # a large jump fails validate_homography: the identity goes into the history, the frame is
    # still stitched and the state still advances (main.py:734-759)
    world = synth.make_world(520, 620, seed=0)
    far = synth.render_frame(world, synth.camera_matrix(0, 64 + 60.0, 520 - FH - 64.0 - 25.0), FH, FW)
    h_before, last_valid, canvas_before = vm.H_old.copy(), vm.last_valid_H.copy(), vm.output_img.copy()
    n_hist = len(vm.homography_history)
    vm.process_frame(far, 3)
    vm.flush()
    out = capsys.readouterr().out
    assert "большое смещение" in out and "Невалидная гомография" in out
    assert np.array_equal(vm.homography_history[-1], np.eye(3)) and len(vm.homography_history) == n_hist + 1
    assert np.array_equal(vm.last_valid_H, last_valid)
    assert np.array_equal(vm.H, vm.H_old) and not np.array_equal(vm.H_old, h_before)
    assert not np.array_equal(vm.output_img, canvas_before)
    assert np.array_equal(vm.des_prev, vm.des_cur)         # state advanced to the rejected frame
    vm.close()


@pytest.mark.gpu
def test_stage_methods(capsys):
    frames, _ = video(3)
    vm = rtvm.VideMosaic(frames[0], detector_type="orb", visualize=False, nfeatures=300)
    vm.process_frame(frames[1], 1)
    ms = vm.match(vm.des_cur, vm.des_prev)
    assert all(ms[i].distance <= ms[i + 1].distance for i in range(len(ms) - 1))
    H = rtvm.VideMosaic.findHomography(vm.kp_cur, vm.kp_prev, ms)
    assert H is not None and H.shape == (3, 3) and H.dtype == np.float64
    assert rtvm.VideMosaic.findHomography(vm.kp_cur, vm.kp_prev, ms[:3]) is None
    c = rtvm.VideMosaic.get_transformed_corners(frames[0], np.array([[1, 0, 2.7], [0, 1, -3.2], [0, 0, 1.0]]))
    assert c.dtype == np.int32 and c.shape == (1, 4, 2)
    assert c[0].tolist() == [[2, -3], [FW + 2, -3], [FW + 2, FH - 4], [2, FH - 4]]
    out = vm.warp(frames[2], vm.H_old)
    assert out.dtype == np.float64 and out.shape == vm.output_img.shape
    vm.close()


@pytest.mark.gpu
@pytest.mark.parametrize("fixed", [False, True])
def test_device_stream_path_matches_oracle(fixed):
    """rtvm_stream_begin / rtvm_process_frame with device-resident frames."""
    import torch
    frames, _ = video(9, seed=4)
    flags = rtvm._ffi.FLAG_WARP_FIXED if fixed else 0
    ctx = rtvm.Context(FH, FW, 2 * FH, int(1.2 * FW), nfeatures=500, flags=flags)
    om = OracleMosaic(frames[0], nfeatures=500, fixed=fixed, glue="explicit")
    dev = [torch.from_numpy(f).cuda() for f in frames]
    flat = torch.full_like(dev[0], 90)
    torch.cuda.synchronize()
    ctx.stream_begin(dev[0].data_ptr(), om.w_offset, om.h_offset)
    seq = [1, 2, 3, "flat", 4, 5, 6, 7, 8]
    for s in seq:
        if s == "flat":
            ctx.process_frame(flat.data_ptr())
            om.process_frame(np.full_like(frames[0], 90))
        else:
            ctx.process_frame(dev[s].data_ptr())
            om.process_frame(frames[s])
        st, nk, nm, H = ctx.frame_status()
        assert st == om.status and nm == om.n_matches
        if st & 3:
            continue
        assert np.array_equal(H, om.H), "frame %s" % s
        assert np.array_equal(ctx.canvas_read(), om.canvas), "frame %s" % s
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("batch,flags,fused", [(1, 0, "0"), (3, 0, "1"), (4, 0, "0"), (6, 0, "1"), (7, 0, "0"), (4, 4, "1"),
                                               (6, 4, "0"), (7, 0, "1")])
def test_batched_stream_matches_oracle(batch, flags, fused, monkeypatch):
    """rtvm_process_batch: same canvas and homography as frame-by-frame processing, including a
    skipped frame in the middle of a batch (later frames come back as pending and are resubmitted)
    and at the end of a batch (the batch queued ahead of it comes back pending as a whole).
    flags 4 = everything on one stream; fused: the column pass of the blur inside the commit kernel, or not."""
    monkeypatch.setenv("RTVM_FUSED_COMMIT", fused)
    import torch
    frames, _ = video(12, seed=6)
    order = [1, 2, 3, 4, 5, "flat", 6, 7, 8, 9, 10, 11]
    seq = [np.full_like(frames[0], 90) if s == "flat" else frames[s] for s in order]
    ctx = rtvm.Context(FH, FW, 2 * FH, int(1.2 * FW), nfeatures=500, flags=flags)
    om = OracleMosaic(frames[0], nfeatures=500, glue="explicit")
    first = torch.from_numpy(frames[0]).cuda()
    dev = [torch.from_numpy(f).cuda() for f in seq]
    torch.cuda.synchronize()
    ctx.stream_begin(first.data_ptr(), om.w_offset, om.h_offset)
    status = ctx.process_video([d.data_ptr() for d in dev], batch=batch)
    ref = []
    for f in seq:
        om.process_frame(f)
        ref.append(om.status)
    assert status.tolist() == ref
    st, nk, nm, H = ctx.frame_status()
    assert np.array_equal(H, om.H)
    assert np.array_equal(ctx.canvas_read(), om.canvas)
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("detector", ["orb", "sift"])
def test_streams_laid_out_by_the_queue_probe_or_as_the_runtime_deals_them(detector, monkeypatch, capfd):
    """A context picks its streams by which hardware queue they sit on (a probe at creation); RTVM_NO_QUEUE_PROBE
    leaves them where the runtime puts them.  Same results either way, host frames through the ingest ring."""
    frames, _ = video(14)
    det = rtvm._ffi.DET_SIFT if detector == "sift" else rtvm._ffi.DET_ORB
    got = []
    for plain in (False, True):
        if plain:
            monkeypatch.setenv("RTVM_NO_QUEUE_PROBE", "1")
        monkeypatch.setenv("RTVM_QUEUE_DEBUG", "1")
        ctx = rtvm.Context(FH, FW, 2 * FH, int(1.2 * FW), nfeatures=400, detector=det)
        st = ctx.process_host_video(frames[0], frames[1:], FH, int(1.2 * FW / 2 - FW / 2), batch=4)
        got.append((st.tolist(), ctx.frame_status()[3], ctx.canvas_read()))
        ctx.close()
        err = capfd.readouterr().err
        assert ("candidate streams, queue classes" in err) == (not plain)
        assert "copy stream is candidate" in err
    assert got[0][0] == got[1][0] == [0] * 13
    assert np.array_equal(got[0][1], got[1][1]) and np.array_equal(got[0][2], got[1][2])


@pytest.mark.gpu
def test_batches_queued_back_to_back():
    """Batches queued without reading any status in between (two in flight on two streams):
    same canvas as frame-by-frame processing; after a skipped frame everything later is pending."""
    import torch
    frames, _ = video(13, seed=8)
    ctx = rtvm.Context(FH, FW, 2 * FH, int(1.2 * FW), nfeatures=500)
    om = OracleMosaic(frames[0], nfeatures=500, glue="explicit")
    dev = [torch.from_numpy(f).cuda() for f in frames]
    flat = torch.full_like(dev[0], 90)
    torch.cuda.synchronize()
    ctx.reserve_batch(4)
    ctx.stream_begin(dev[0].data_ptr(), om.w_offset, om.h_offset)
    for b in range(3):
        ctx.process_batch([dev[1 + 4 * b + j].data_ptr() for j in range(4)])
    done1, st1, _, _ = ctx.batch_status(4, back=1)
    done0, st0, _, _ = ctx.batch_status(4, back=0)
    assert done1 == 4 and done0 == 4 and not st1.any() and not st0.any()
    for f in frames[1:13]:
        om.process_frame(f)
    st, nk, nm, H = ctx.frame_status()
    assert np.array_equal(H, om.H)
    assert np.array_equal(ctx.canvas_read(), om.canvas)
    # a skipped frame at the end of a batch voids the two batches queued behind it
    ctx.stream_begin(dev[0].data_ptr(), om.w_offset, om.h_offset)
    ctx.process_batch([dev[1].data_ptr(), dev[2].data_ptr(), flat.data_ptr()])
    ctx.process_batch([dev[3].data_ptr(), dev[4].data_ptr()])
    ctx.process_batch([dev[5].data_ptr(), dev[6].data_ptr()])
    done, stl, _, _ = ctx.batch_status(2, back=0)
    assert done == 0 and stl.tolist() == [8, 8]
    # the status read ended the overlap: the next batch is matched against frame 2
    ctx.process_batch([dev[3].data_ptr(), dev[4].data_ptr()])
    done, stl, _, _ = ctx.batch_status(2)
    assert done == 2 and stl.tolist() == [0, 0]
    om2 = OracleMosaic(frames[0], nfeatures=500, glue="explicit")
    for f in (frames[1], frames[2], np.full_like(frames[0], 90), frames[3], frames[4]):
        om2.process_frame(f)
    st, nk, nm, H = ctx.frame_status()
    assert np.array_equal(H, om2.H)
    assert np.array_equal(ctx.canvas_read(), om2.canvas)
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("fused", ["0", "1"])
def test_batch_with_pixels_that_blend_to_zero(fused, monkeypatch):
    """Two pixels of value 1 blend to 0 (the blurred weights add up to slightly less than 1), which
    clears the canvas mask there.  The batched path predicts the canvas masks of a batch ahead of
    time on the assumption that this does not happen; when it does, the later frames of the batch
    come back pending and are processed again: the result stays the frame-by-frame one."""
    monkeypatch.setenv("RTVM_FUSED_COMMIT", fused)   # both ways of ending the canvas half rebuild the pixels of a bad piece
    import torch
    frames, _ = video(11, seed=9)
    frames = [f.copy() for f in frames]
    for f in frames:
        f[90:150, 200:280] = 1
    ctx = rtvm.Context(FH, FW, 2 * FH, int(1.2 * FW), nfeatures=500)
    om = OracleMosaic(frames[0], nfeatures=500, glue="explicit")
    dev = [torch.from_numpy(f).cuda() for f in frames]
    torch.cuda.synchronize()
    ctx.reserve_batch(5)
    ctx.stream_begin(dev[0].data_ptr(), om.w_offset, om.h_offset)
    ctx.process_batch([d.data_ptr() for d in dev[1:6]])
    done, st, _, _ = ctx.batch_status(5)
    assert 1 <= done < 5 and st[:done].tolist() == [0] * done and st[done:].tolist() == [8] * (5 - done)
    status = ctx.process_video([d.data_ptr() for d in dev[1 + done:]], batch=5)
    ref = []
    for f in frames[1:]:
        om.process_frame(f)
        ref.append(om.status)
    assert [0] * done + status.tolist() == ref
    st, nk, nm, H = ctx.frame_status()
    assert np.array_equal(H, om.H)
    canvas = ctx.canvas_read()
    assert np.array_equal(canvas, om.canvas)
    assert (canvas[om.canvas.any(axis=2)].size > 0)
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("fused", ["0", "1"])
def test_batch_retry_queues_only_the_canvas_half_again(fused, monkeypatch):
    """rtvm_batch_retry: frames left pending because a blend cleared a canvas pixel keep their
    detection, matching and RANSAC results; their canvas half alone is queued again, for them and
    for the batch that was in flight behind them."""
    monkeypatch.setenv("RTVM_FUSED_COMMIT", fused)
    import torch
    F = rtvm._ffi
    frames, _ = video(9, seed=9)
    frames = [f.copy() for f in frames]
    for f in frames:
        f[90:150, 200:280] = 1
    ctx = rtvm.Context(FH, FW, 2 * FH, int(1.2 * FW), nfeatures=500)
    om = OracleMosaic(frames[0], nfeatures=500, glue="explicit")
    dev = [torch.from_numpy(f).cuda() for f in frames]
    torch.cuda.synchronize()
    ctx.reserve_batch(4)
    ctx.stream_begin(dev[0].data_ptr(), om.w_offset, om.h_offset)
    ctx.process_batch([d.data_ptr() for d in dev[1:5]])
    t1 = ctx.batch_ticket()
    ctx.process_batch([d.data_ptr() for d in dev[5:9]])
    t2 = ctx.batch_ticket()
    assert t2 == t1 + 1
    done, reason, st, _, _ = ctx.batch_status_of(t1, 4)
    assert 1 <= done < 4 and reason == F.PENDING_MASK and st[done:].tolist() == [8] * (4 - done)
    assert ctx.batch_cleared(t1) == done - 1   # the last frame that went through is the one that cleared a pixel
    done2, reason2, st2, _, _ = ctx.batch_status_of(t2, 4)
    assert done2 == 0 and reason2 == F.PENDING_VOID and st2.tolist() == [8] * 4
    assert ctx.batch_cleared(t2) == -1
    retries = 0
    for ticket in (t1, t2):
        first = done if ticket == t1 else 0
        while first < 4:
            t = ctx.batch_retry(ticket, first, 4 - first)
            retries += 1
            d, r, s_, _, _ = ctx.batch_status_of(t, 4 - first)
            assert d >= 1 and s_[:d].tolist() == [0] * d and (d == 4 - first or r == F.PENDING_MASK)
            first += d
    assert retries >= 3
    for f in frames[1:]:
        om.process_frame(f)
        assert om.status == 0
    st, nk, nm, H = ctx.frame_status()
    assert np.array_equal(H, om.H)
    assert np.array_equal(ctx.canvas_read(), om.canvas)
    with pytest.raises(RuntimeError):
        ctx.batch_retry(t2 + 100, 0, 1)
    with pytest.raises(RuntimeError):
        ctx.batch_retry(t1, 3, 2)
    ctx.close()


@pytest.mark.gpu
def test_detection_queued_ahead_of_the_canvas_half():
    """rtvm_batch_detect / rtvm_batch_retry / rtvm_batch_release: four batches detected ahead, their
    canvas halves queued afterwards in frame order; a fifth detection needs a released slot set."""
    import torch
    frames, _ = video(13, seed=12)
    ctx = rtvm.Context(FH, FW, 2 * FH, int(1.2 * FW), nfeatures=500)
    om = OracleMosaic(frames[0], nfeatures=500, glue="explicit")
    dev = [torch.from_numpy(f).cuda() for f in frames]
    torch.cuda.synchronize()
    ctx.reserve_batch(3)
    ctx.stream_begin(dev[0].data_ptr(), om.w_offset, om.h_offset)
    tickets = [ctx.batch_detect([d.data_ptr() for d in dev[1 + 3 * b:4 + 3 * b]]) for b in range(4)]
    with pytest.raises(RuntimeError):
        ctx.batch_detect([dev[1].data_ptr()])
    for t in tickets:
        bt = ctx.batch_retry(t, 0, 3)
        done, reason, st, nm, nk = ctx.batch_status_of(bt, 3)
        assert done == 3 and reason == 0 and st.tolist() == [0, 0, 0] and (nm > 20).all()
        ctx.batch_release(t)
    for f in frames[1:13]:
        om.process_frame(f)
    st, nk, nm, H = ctx.frame_status()
    assert np.array_equal(H, om.H)
    assert np.array_equal(ctx.canvas_read(), om.canvas)
    # released: detection can go on, and the plain batch call still works after it
    t = ctx.batch_detect([dev[1].data_ptr()])
    ctx.batch_release(t)
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("slots", [12, 15])
def test_host_frames_through_the_ingest_ring_give_the_same_mosaic(slots):
    """rtvm_ingest_*: frames in host memory, copied through pinned slots on a stream of their own while
    earlier batches are stitched, against the oracle frame by frame (more frames than ring slots per
    batch set, so slots are reused; with five batches of slots the next batch is copied ahead in the
    background)."""
    frames, _ = video(23, seed=14)
    ctx = rtvm.Context(FH, FW, 2 * FH, int(1.2 * FW), nfeatures=500)
    om = OracleMosaic(frames[0], nfeatures=500, glue="explicit")
    ctx._ring = rtvm._ffi.HostFrameRing(ctx, slots)   # four or five batches of 3
    st = ctx.process_host_video(frames[0], frames[1:], om.w_offset, om.h_offset, batch=3)
    want = []
    for f in frames[1:]:
        om.process_frame(f)
        want.append(om.status)
    assert st.tolist() == want
    _, _, _, H = ctx.frame_status()
    assert np.array_equal(H, om.H)
    assert np.array_equal(ctx.canvas_read(), om.canvas)
    # slots hand out pinned host memory of the frame's size; a second ring is refused
    assert ctx.ingest_host_slot(0).shape == (FH, FW, 3)
    with pytest.raises(RuntimeError):
        ctx.ingest_reserve(4)
    with pytest.raises(RuntimeError):
        ctx.ingest_submit(99)
    ctx.close()


@pytest.mark.gpu
def test_dropin_with_the_default_sift_detector_matches_oracle(capsys):
    """VideMosaic(first_image) with the reference's default detector_type="sift": SIFT features, knn-2 + ratio
    matches (main.py:688-692), RANSAC, warp + blend, frame by frame against the oracle."""
    frames, _ = video(6, seed=15)
    vm = rtvm.VideMosaic(frames[0], visualize=False, show_intermediate=False, nfeatures=400)
    assert vm.detector_type == "sift"
    om = OracleMosaic(frames[0], nfeatures=400, detector="sift")
    for t, f in enumerate(frames[1:], 1):
        vm.process_frame(f, t)
        om.process_frame(f)
        assert om.status == 0 and om.n_matches > 30
        assert len(vm.matches) == om.n_matches
        assert [(m.queryIdx, m.trainIdx) for m in vm.matches] == list(zip(om.matches[0].tolist(), om.matches[1].tolist()))
        assert np.array_equal(vm.H, om.H)
    assert np.array_equal(vm.des_prev, om.feat_prev["desc"]) and vm.des_prev.shape[1] == 128
    assert len(vm.kp_prev) == len(om.feat_prev["xy"])
    assert np.array_equal(vm.output_img, om.canvas.astype(np.float64))
    vm.close()


@pytest.mark.gpu
@pytest.mark.parametrize("detector,workers,display", [("sift", 3, True), ("sift", 1, True), ("orb", 4, True),
                                                      ("orb", 4, False), ("sift", 2, False)])
def test_process_frames_equals_frame_by_frame(capsys, detector, workers, display):
    """process_frames (SIFT: detection, matching and RANSAC of several frames on worker threads ahead of the
    sequential part) leaves the same mosaic, homographies, features, matches and printed warnings as one
    process_frame call per frame, also across skipped frames (a black one, a repeated one that follows it)."""
    frames, _ = video(9, seed=16)
    frames = list(frames)
    frames.insert(4, np.zeros_like(frames[0]))        # no features: skipped, the next frame matches frame 3
    frames.insert(7, np.zeros_like(frames[0]))
    frames.insert(8, np.zeros_like(frames[0]))        # two skipped frames in a row
    # display: every call completes before it returns (pipeline_batch=0), process_frames runs detection on worker
    # threads; otherwise process_frame itself only enqueues and process_frames is a loop over it
    kw = dict(detector_type=detector, visualize=display, show_intermediate=display, nfeatures=300,
              pipeline_batch=0 if display else 32)
    a = rtvm.VideMosaic(frames[0], **kw)
    Hs = []
    for t, f in enumerate(frames[1:], 1):
        a.process_frame(f, t)
        Hs.append(None if a.H is None else a.H.copy())
    out_a = capsys.readouterr().out
    b = rtvm.VideMosaic(frames[0], **kw)
    b.process_frames(frames[1:5], 1, workers=workers)
    assert np.array_equal(b.H, Hs[3])
    b.process_frames(frames[5:], 5, workers=workers)
    b.flush()                                         # headless: the warnings of frames still in flight
    out_b = capsys.readouterr().out
    assert out_a == out_b and out_a.count("пропуск кадра") == 3
    assert np.array_equal(a.H, b.H) and np.array_equal(a.H_old, b.H_old)
    assert len(a.homography_history) == len(b.homography_history)
    assert all(np.array_equal(x, y) for x, y in zip(a.homography_history, b.homography_history))
    assert np.array_equal(a.des_prev, b.des_prev) and np.array_equal(a.des_cur, b.des_cur)
    assert [(m.queryIdx, m.trainIdx, m.distance) for m in a.matches] == [(m.queryIdx, m.trainIdx, m.distance) for m in b.matches]
    assert np.array_equal(a.output_img_u8(), b.output_img_u8())
    a.close()
    b.close()


@pytest.mark.gpu
@pytest.mark.parametrize("detector,nf", [("orb", 1000), ("sift", 700)])
def test_process_frames_at_720p_equals_frame_by_frame(detector, nf):
    """The bench's video at full size: 24 frames through process_frames in two calls against process_frame."""
    import torch
    fh, fw = 720, 1280
    Gs, wh, ww = rtvm.synth.bounded_path(25, fh, fw, seed=2)
    world = rtvm.synth.make_world(wh, ww, seed=0)
    frames = list(rtvm.synth.render_video_torch(world, Gs, fh, fw, torch.device("cuda", 0)).cpu().numpy())
    kw = dict(detector_type=detector, nfeatures=nf, pipeline_batch=0)   # frame-synchronous: detection on worker threads
    a = rtvm.VideMosaic(frames[0], **kw)
    for t, f in enumerate(frames[1:], 1):
        a.process_frame(f, t)
    b = rtvm.VideMosaic(frames[0], **kw)
    b.process_frames(frames[1:14], 1, workers=4)
    b.process_frames(frames[14:], 14, workers=3)
    assert np.array_equal(a.H, b.H) and len(a.homography_history) == len(b.homography_history)
    assert np.array_equal(a.des_prev, b.des_prev) and np.array_equal(a.des_cur, b.des_cur)
    assert [(m.queryIdx, m.trainIdx, m.distance) for m in a.matches] == [(m.queryIdx, m.trainIdx, m.distance) for m in b.matches]
    assert np.array_equal(a.output_img_u8(), b.output_img_u8())
    a.close()
    b.close()


@pytest.mark.gpu
@pytest.mark.parametrize("fh,fw,ch,cw,nf,batch,n", [(1080, 1920, 2160, 2304, 1000, 4, 6),
                                                    (2160, 3840, 4096, 16384, 4000, 3, 4)])
def test_full_size_configs_batched_match_oracle_and_frame_by_frame(fh, fw, ch, cw, nf, batch, n):
    """The 1080p stream shape and the 4K / 16384-wide-canvas configuration at their own sizes: the batched device
    pipeline (detection ahead, predicted masks, pieces) against the CPU restatement frame by frame -- statuses,
    match count, homography and every canvas pixel -- and against frame-by-frame processing on the device."""
    import torch
    Gs, wh, ww = rtvm.synth.bounded_path(n + 1, fh, fw, seed=21)
    world = rtvm.synth.make_world(wh, ww, seed=3, n_shapes=4000)
    dev = torch.device("cuda", 0)
    frames = rtvm.synth.render_video_torch(world, Gs, fh, fw, dev)
    torch.cuda.synchronize()
    host = frames.cpu().numpy()
    stride = fh * fw * 3
    base = frames.data_ptr()
    row_off, col_off = ch - fh, int(cw / 2 - fw / 2)
    a = rtvm.Context(fh, fw, ch, cw, nfeatures=nf)
    a.stream_begin(base, row_off, col_off)
    st_a = []
    for j in range(1, n + 1):
        a.process_frame(base + j * stride)
        st_a.append(a.frame_status()[0])
    _, _, nm_a, H_a = a.frame_status()
    canvas_a = a.canvas_read()
    a.close()
    b = rtvm.Context(fh, fw, ch, cw, nfeatures=nf)
    b.stream_begin(base, row_off, col_off)
    st_b = b.process_video([base + j * stride for j in range(1, n + 1)], batch=batch)
    _, _, nm_b, H_b = b.frame_status()
    canvas_b = b.canvas_read()
    b.close()
    assert st_a == st_b.tolist() and st_a == [0] * n and nm_a == nm_b and nm_a > 100
    assert np.array_equal(H_a, H_b)
    assert np.array_equal(canvas_a, canvas_b)
    assert canvas_a[row_off:row_off + 8, col_off:col_off + 8].any()
    del canvas_a
    om = OracleMosaic(host[0], nfeatures=nf, canvas_size=(ch, cw), glue="explicit")
    for f in host[1:]:
        om.process_frame(f)
        assert om.status == 0
    assert nm_b == om.n_matches
    assert np.array_equal(H_b, om.H)
    assert np.array_equal(canvas_b, om.canvas)
