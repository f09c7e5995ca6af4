"""VideMosaic's pipeline (the default for every combination of the display flags): process_frame only takes
the frame, the work follows in batches, and reading any result brings the object to the state the
reference has after the last call.  Checked against the oracle and against the frame-synchronous class
(pipeline_batch=0)."""
import numpy as np
import pytest

import reject_cases as rc
import rtvm
from rtvm import synth
from oracle_mosaic import OracleMosaic

FH, FW = 240, 320
HEADLESS = dict(detector_type="orb", visualize=False, show_intermediate=False)


def state_of(vm):
    return dict(H=vm.H, H_old=vm.H_old, last_valid_H=vm.last_valid_H, history=list(vm.homography_history),
                des_prev=vm.des_prev, des_cur=vm.des_cur, kp_cur=[(k.pt, k.angle, k.octave) for k in vm.kp_cur],
                kp_prev=[(k.pt, k.response) for k in vm.kp_prev],
                matches=[(m.queryIdx, m.trainIdx, m.distance) for m in vm.matches], canvas=vm.output_img_u8())


def assert_same_state(a, b):
    assert sorted(a) == sorted(b)
    for k in a:
        if k == "history":
            assert len(a[k]) == len(b[k]) and all(np.array_equal(x, y) for x, y in zip(a[k], b[k]))
        elif isinstance(a[k], list):
            assert a[k] == b[k], k
        elif a[k] is None:
            assert b[k] is None
        else:
            assert np.array_equal(a[k], b[k]), k


def tricky_video():
    """Rejected homographies (every rule), frames without features in the middle of a batch, at its end and two
    in a row, and patches of value 1 whose blends clear canvas pixels (mispredicted masks)."""
    frames, _ = rc.sequence()
    frames = [f.copy() for f in frames]
    for f in frames:
        f[90:150, 200:280] = 1
    flat = np.full_like(frames[0], 90)
    black = np.zeros_like(frames[0])
    for pos, f in ((4, flat), (8, black), (13, flat), (14, black), (20, flat)):
        frames.insert(pos, f)
    return frames


@pytest.mark.gpu
@pytest.mark.parametrize("detector,batch,display", [("orb", 1, (False, False)), ("orb", 4, (True, True)),
                                                    ("orb", 8, (False, True)), ("orb", 32, (True, False)),
                                                    ("orb", 32, (False, False)), ("sift", 1, (False, False)),
                                                    ("sift", 5, (True, True)), ("sift", 16, (False, False))])
def test_headless_pipeline_equals_frame_by_frame_and_oracle(detector, batch, display, capsys):
    """display = (visualize, show_intermediate): accepted, and without effect on results or on the mode."""
    frames = tricky_video()
    sync = rtvm.VideMosaic(frames[0], detector_type=detector, nfeatures=500, pipeline_batch=0)
    assert not sync._deferred
    om = OracleMosaic(frames[0], nfeatures=500, glue="numpy", detector=detector)
    for t, f in enumerate(frames[1:], 1):
        sync.process_frame(f, t)
        om.process_frame(f)
    out_sync = capsys.readouterr().out
    assert out_sync.count("пропуск кадра") == 5 and out_sync.count("Невалидная гомография") >= (5 if detector == "orb" else 1)
    assert np.array_equal(sync.H, om.H) and np.array_equal(sync.output_img_u8(), om.canvas)
    want = state_of(sync)

    kw = dict(detector_type=detector, visualize=display[0], show_intermediate=display[1])
    vm = rtvm.VideMosaic(frames[0], nfeatures=500, pipeline_batch=batch, **kw)
    assert vm._deferred and vm._ctx.detector == (rtvm._ffi.DET_SIFT if detector == "sift" else rtvm._ffi.DET_ORB)
    for t, f in enumerate(frames[1:], 1):
        assert vm.process_frame(f, t) is None
    assert vm.frame_cur is frames[-1]
    got = state_of(vm)
    out = capsys.readouterr().out
    assert out == out_sync                      # the same warnings in the same order
    assert_same_state(got, want)
    stats = vm._pipe.stats
    assert stats["frames"] == len(frames) - 1 and stats["frames_detected_again"] > 0
    if batch >= 4 and detector == "orb":
        assert stats["pieces_with_pending_frames"] > 0
    # the object goes on from there, and reads in the middle of a video change nothing
    more = [synth.render_frame(rc.world(), rc.view(dx=5 + 2 * i, dy=-4 - i), FH, FW) for i in range(1, 6)]
    for i, f in enumerate(more):
        sync.process_frame(f, 100 + i)
        vm.process_frame(f, 100 + i)
        if i == 2:
            assert np.array_equal(vm.H, sync.H) and len(vm.matches) == len(sync.matches)
    assert_same_state(state_of(vm), state_of(sync))
    sync.close()
    vm.close()


@pytest.mark.gpu
def test_headless_attributes_are_live(capsys):
    """The public knobs of the reference (thresholds, history size, stabilisation switch, H_old) are read when a
    frame's rules are applied; setting one first settles the frames handed in before."""
    frames, _ = synth.make_video(FH, FW, 12, world_size=(520, 620), seed=3)
    a = rtvm.VideMosaic(frames[0], detector_type="orb", visualize=True, nfeatures=400, pipeline_batch=0)
    b = rtvm.VideMosaic(frames[0], nfeatures=400, pipeline_batch=4, **HEADLESS)
    for vm in (a, b):
        for t in range(1, 5):
            vm.process_frame(frames[t], t)
        vm.translation_threshold = 1.0          # every later frame is "shaky"
        vm.history_size = 3
        for t in range(5, 9):
            vm.process_frame(frames[t], t)
        vm.stabilization_enabled = False
        vm.translation_threshold = 50
        shifted = vm.H_old.copy()
        shifted[0, 2] += 7.0
        vm.H_old = shifted
        for t in range(9, 12):
            vm.process_frame(frames[t], t)
    out = capsys.readouterr().out
    assert out.count("большое смещение") == 8          # frames 5..8, for both objects
    assert len(a.homography_history) == len(b.homography_history) == 4   # one pop per append (main.py:819-820)
    assert_same_state(state_of(b), state_of(a))
    a.close()
    b.close()


@pytest.mark.gpu
def test_headless_at_720p_matches_oracle_through_a_mask_retry_episode():
    import torch
    from test_headline import bench_video
    fh, fw, ch, cw, nf, n = 720, 1280, 1440, 1536, 1000, 96
    _, host = bench_video(n, fh, fw, torch.device("cuda", 0))
    vm = rtvm.VideMosaic(host[0], nfeatures=nf, canvas_size=(ch, cw), **HEADLESS)
    for t in range(1, n + 1):
        vm.process_frame(host[t], t)
    H, canvas, matches = vm.H, vm.output_img_u8(), [(m.queryIdx, m.trainIdx) for m in vm.matches]
    stats = dict(vm._pipe.stats)
    vm.close()
    assert stats["pieces_with_pending_frames"] > 0 and stats["batches"] == 3
    om = OracleMosaic(host[0], nfeatures=nf, canvas_size=(ch, cw), glue="numpy")
    for f in host[1:]:
        om.process_frame(f)
        assert om.status == 0
    assert np.array_equal(H, om.H)
    assert matches == list(zip(om.matches[0].tolist(), om.matches[1].tolist()))
    assert np.array_equal(canvas, om.canvas)


@pytest.mark.gpu
@pytest.mark.parametrize("batch", [1, 4, 7])
def test_device_stream_with_the_sift_detector_matches_oracle(batch):
    """Context(detector=DET_SIFT): SIFT features, knn-2 + ratio + sort and RANSAC on the device for the frames of a
    batch, the reference's rules in k_batch_glue, against the oracle; with a frame without features in mid-batch."""
    import torch
    frames, _ = synth.make_video(FH, FW, 12, world_size=(520, 620), seed=6)
    order = [1, 2, 3, 4, 5, "flat", 6, 7, 8, 9, 10, 11]
    seq = [np.full_like(frames[0], 90) if s == "flat" else frames[s] for s in order]
    ctx = rtvm.Context(FH, FW, 2 * FH, int(1.2 * FW), detector=rtvm._ffi.DET_SIFT, nfeatures=400)
    om = OracleMosaic(frames[0], nfeatures=400, glue="explicit", detector="sift")
    first = torch.from_numpy(frames[0]).cuda()
    dev = [torch.from_numpy(f).cuda() for f in seq]
    torch.cuda.synchronize()
    ctx.stream_begin(first.data_ptr(), om.w_offset, om.h_offset)
    f0 = ctx.get_features(1)
    for k in ("xy", "size", "angle", "response", "octave", "desc"):
        assert np.array_equal(f0[k], om.feat_prev[k]), k
    if batch == 1:
        status = []
        for d in dev:
            ctx.process_frame(d.data_ptr())
            status.append(ctx.frame_status()[0])
    else:
        status = ctx.process_video([d.data_ptr() for d in dev], batch=batch).tolist()
    ref = []
    for f in seq:
        om.process_frame(f)
        ref.append(om.status)
    assert status == ref and ref.count(1) == 1
    st, nk, nm, H = ctx.frame_status()
    assert nm == om.n_matches and nm > 30
    assert np.array_equal(H, om.H)
    assert np.array_equal(ctx.canvas_read(), om.canvas)
    fp = ctx.get_features(1)
    assert np.array_equal(fp["desc"], om.feat_prev["desc"]) and fp["desc"].dtype == np.float32
    with pytest.raises(RuntimeError):
        ctx.orb(frames[0])
    ctx.close()


@pytest.mark.gpu
def test_an_exception_in_the_callers_rules_surfaces_from_process_frame_or_flush():
    """The rules run inside the library's call-back; what they raise is raised again by the call that was in
    progress (process_frame or a read), not swallowed."""
    frames, _ = synth.make_video(FH, FW, 8, world_size=(520, 620), seed=3)
    vm = rtvm.VideMosaic(frames[0], detector_type="orb", nfeatures=300, pipeline_batch=4)
    calls = []

    def picky(H):
        calls.append(1)
        if len(calls) == 3:
            raise ValueError("no thanks")
        return True
    vm.validate_homography = picky
    with pytest.raises(ValueError, match="no thanks"):
        for t in range(1, 8):
            vm.process_frame(frames[t], t)
        vm.flush()
    assert len(calls) == 3
    with pytest.raises(rtvm._ffi.RtvmError) as e:       # the frames in flight are in an undefined state: nothing more is taken
        vm.process_frame(frames[1], 8)
    assert e.value.code == rtvm._ffi.E_STATE and "close" in str(e.value)
    vm.close()


@pytest.mark.gpu
def test_process_video_through_a_frame_source_with_events_and_a_failing_source():
    """rtvm_process_video_source: the frames handed out batch by batch by a Python function, the canvas halves'
    outcomes as events; a source that fails ends the call with its own exception."""
    import torch
    frames, _ = synth.make_video(FH, FW, 21, world_size=(520, 620), seed=5)
    frames = frames[:21]
    dev = [torch.from_numpy(f).cuda() for f in frames]
    torch.cuda.synchronize()
    ptrs = [d.data_ptr() for d in dev[1:]]

    class Source:
        def __init__(self, fail_at=None):
            self.asked, self.fail_at = [], fail_at

        def __len__(self):
            return len(ptrs)

        def acquire(self, first, k):
            self.asked.append((first, k))
            if self.fail_at is not None and first >= self.fail_at:
                raise KeyError("frame %d is gone" % first)
            return ptrs[first:first + k]

    a = rtvm.Context(FH, FW, 2 * FH, int(1.2 * FW), nfeatures=400)
    a.stream_begin(dev[0].data_ptr(), FH, int(1.2 * FW / 2 - FW / 2))
    st_a = a.process_video(ptrs, batch=8)
    want = a.canvas_read()
    a.close()
    b = rtvm.Context(FH, FW, 2 * FH, int(1.2 * FW), nfeatures=400)
    b.stream_begin(dev[0].data_ptr(), FH, int(1.2 * FW / 2 - FW / 2))
    src, trace = Source(), []
    st_b = b.process_video(src, batch=8, trace=trace)
    assert st_b.tolist() == st_a.tolist() == [0] * 20 and np.array_equal(b.canvas_read(), want)
    assert src.asked == [(0, 8), (8, 8), (16, 4)]
    # every frame consumed once; a piece that leaves frames pending says why (here: a blend that cleared a pixel)
    assert sum(e[3] for e in trace) == 20 and all((e[4] == 0) == (e[3] == e[2]) and e[4] in (0, rtvm._ffi.PENDING_MASK) for e in trace), trace
    assert all(t1[0] >= t0[0] for t0, t1 in zip(trace, trace[1:]))
    assert b.video_stats["frames"] == 20 and b.video_stats["canvas_frames"] == 20
    b.stream_begin(dev[0].data_ptr(), FH, int(1.2 * FW / 2 - FW / 2))
    with pytest.raises(KeyError, match="frame 8 is gone"):
        b.process_video(Source(fail_at=8), batch=8)
    # no slot set stays taken after the failure: the context goes on with the next video
    b.stream_begin(dev[0].data_ptr(), FH, int(1.2 * FW / 2 - FW / 2))
    assert b.process_video(ptrs, batch=8).tolist() == [0] * 20 and np.array_equal(b.canvas_read(), want)
    b.close()


@pytest.mark.gpu
def test_pipe_entry_points_refuse_misuse():
    F = rtvm._ffi
    frames, _ = synth.make_video(FH, FW, 3, world_size=(520, 620), seed=5)
    ctx = rtvm.Context(FH, FW, 2 * FH, int(1.2 * FW), nfeatures=300)
    noop = F.GLUE_FN(lambda *a: 0)
    with pytest.raises(F.RtvmError) as e:       # no stream yet
        ctx.pipe_open(4, 2, noop)
    assert e.value.code == F.E_STATE
    ctx.first_frame(frames[0], FH, 32)
    inside = []

    def glue(user, index, nm, found, H_rel, H_abs):
        inside.append(F.lib().rtvm_pipe_flush(ctx._h))     # from inside the rules: refused, not a deadlock
        return 0
    fn = F.GLUE_FN(glue)
    ctx.pipe_open(2, 1, fn)
    with pytest.raises(F.RtvmError) as e:       # one pipe per context
        ctx.pipe_open(2, 1, fn)
    assert e.value.code == F.E_STATE
    for f in frames[1:]:
        assert ctx.pipe_put(np.ascontiguousarray(f)) == F.OK
    assert ctx.pipe_flush() == F.OK and not ctx.pipe_busy()
    assert inside and all(rc == F.E_STATE for rc in inside)
    assert ctx.pipe_stats()["frames"] == 2
    ctx.pipe_close()
    ctx.pipe_open(2, 1, fn)                     # and again after closing; the ring is kept
    ctx.close()
