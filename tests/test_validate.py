"""validate_homography (reference main.py:761-801): the reject branch, rule by rule, on the CPU
restatement and through every GPU path (stage call, drop-in class, batched device stream)."""
import numpy as np
import pytest

import oracle_ffi as orc
import reject_cases as rc
import rtvm
from oracle_mosaic import OracleMosaic, INVALID

SPECIAL = {
    "identity": np.eye(3),
    "nan": np.array([[1, 0, 0], [0, np.nan, 0], [0, 0, 1.0]]),
    "inf": np.array([[1, 0, np.inf], [0, 1, 0], [0, 0, 1.0]]),
    # det < 0: sqrt gives NaN, no comparison fails, the reference accepts (main.py:784, 792)
    "mirror": np.array([[-1, 0, 3], [0, 1, 2], [0, 0, 1.0]]),
    "swap": np.array([[0, 1, 0], [1, 0, 0], [0, 0, 1.0]]),
    # singular 2x2 blocks: determinant 0, scale 0, rejected
    "zero_column": np.array([[0, 1, 0], [0, 1, 0], [0, 0, 1.0]]),
    "zero_block": np.array([[0, 0, 1], [0, 0, 1], [0, 0, 1.0]]),
    "rank_one": np.array([[1, 2, 0], [2, 4, 0], [0, 0, 1.0]]),
    "translation_at_threshold": np.array([[1, 0, 30], [0, 1, 40], [0, 0, 1.0]]),       # exactly 50: passes
    "translation_above": np.array([[1, 0, 30], [0, 1, 40.001], [0, 0, 1.0]]),
    "scale_low": np.array([[0.69, 0, 0], [0, 0.69, 0], [0, 0, 1.0]]),
    "scale_high": np.array([[1.31, 0, 0], [0, 1.31, 0], [0, 0, 1.0]]),
    "scale_ok": np.array([[1.29, 0, 0], [0, 1.29, 0], [0, 0, 1.0]]),
    "perspective_x": np.array([[1, 0, 0], [0, 1, 0], [0.0011, 0, 1.0]]),
    "perspective_y": np.array([[1, 0, 0], [0, 1, 0], [0, -0.0011, 1.0]]),
    "perspective_ok": np.array([[1, 0, 0], [0, 1, 0], [0.001, -0.001, 1.0]]),
    "pivot_second_row": np.array([[0.1, 1.0, 0], [-1.0, 0.2, 0], [0, 0, 1.0]]),
}


def _validators():
    a = OracleMosaic.__new__(OracleMosaic)
    b = OracleMosaic.__new__(OracleMosaic)
    a.glue, b.glue = "numpy", "explicit"
    return a, b


def test_special_matrices_numpy_and_explicit_forms_agree():
    a, b = _validators()
    want = {"identity": True, "nan": False, "inf": False, "mirror": True, "swap": True, "zero_column": False,
            "zero_block": False, "rank_one": False, "translation_at_threshold": True, "translation_above": False,
            "scale_low": False, "scale_high": False, "scale_ok": True, "perspective_x": False,
            "perspective_y": False, "perspective_ok": True, "pivot_second_row": True}
    for name, H in SPECIAL.items():
        assert a.validate(H) == b.validate(H) == want[name], name


def test_rejection_sequence_breaks_one_rule_at_a_time():
    """The test video: every labelled frame is rejected by the oracle for exactly its rule; the
    frame after it is accepted again (the state advanced to the rejected frame)."""
    frames, labels = rc.sequence()
    om = OracleMosaic(frames[0], nfeatures=500, glue="explicit")
    seen = set()
    for f, lab in zip(frames[1:], labels):
        prev, hist_before, last_valid = om.feat_prev, len(om.history), om.last_valid_H.copy()
        om.process_frame(f)
        assert not om.status & 3
        if lab is None:
            continue
        qi, ti, _ = om.matches
        H, _ = orc.find_homography(om.feat_cur["xy"][qi], prev["xy"][ti], 2.0)
        rules = rc.broken_rules(H)
        assert om.status == INVALID and sum(rules) == 1 and rules[rc.CASES[lab][0]], (lab, rules)
        assert np.array_equal(om.history[-1], np.eye(3)) and len(om.history) == min(5, hist_before + 1)
        assert np.array_equal(om.last_valid_H, last_valid)
        assert om.feat_prev is om.feat_cur
        seen.add(rc.CASES[lab][0])
    assert seen == {0, 1, 2}


@pytest.mark.gpu
def test_device_validate_matches_oracle(ctx_small):
    a, b = _validators()
    for name, H in SPECIAL.items():
        assert ctx_small.validate_homography(H) == b.validate(H), name
    rng = np.random.default_rng(5)
    n_rejected = 0
    for i in range(300):
        H = np.eye(3) + rng.normal(0, 0.25, (3, 3))
        H[2, :2] = rng.normal(0, 8e-4, 2)
        H[:2, 2] = rng.normal(0, 30, 2)
        if i % 7 == 0:
            H[:2, :2] = H[:2, :2][::-1]          # the larger pivot in the second row, determinant flipped
        want = b.validate(H)
        assert want == a.validate(H)
        assert ctx_small.validate_homography(H) == want, H
        n_rejected += not want
    assert 50 < n_rejected < 250
    # the thresholds are arguments
    assert ctx_small.validate_homography(SPECIAL["translation_above"], 60.0, 0.3)
    assert not ctx_small.validate_homography(SPECIAL["scale_ok"], 50.0, 0.2)


@pytest.mark.gpu
@pytest.mark.parametrize("display,pipeline_batch", [(True, 0), (False, 0), (True, 32), (False, 32)])
def test_rejected_frames_through_the_dropin_match_oracle(capsys, display, pipeline_batch):
    """pipeline_batch=0: every call completes before it returns; otherwise the frame is only taken and flush()
    settles it, whatever the display flags (the reference's defaults: both on) say."""
    frames, labels = rc.sequence()
    vm = rtvm.VideMosaic(frames[0], detector_type="orb", visualize=display, show_intermediate=display, nfeatures=500,
                         pipeline_batch=pipeline_batch)
    assert vm._deferred == (pipeline_batch > 0)
    om = OracleMosaic(frames[0], nfeatures=500, glue="numpy")
    texts = {0: "большое смещение", 1: "большое изменение масштаба", 2: "сильные перспективные искажения"}
    for t, (f, lab) in enumerate(zip(frames[1:], labels), 1):
        vm.process_frame(f, t)
        vm.flush()
        om.process_frame(f)
        out = capsys.readouterr().out
        assert ("Невалидная гомография" in out) == bool(om.status & INVALID), "frame %d" % t
        if lab is not None:
            assert om.status == INVALID and texts[rc.CASES[lab][0]] in out
            assert np.array_equal(vm.homography_history[-1], np.eye(3))
        assert np.array_equal(vm.H, om.H) and np.array_equal(vm.H_old, om.H_old)
        assert np.array_equal(vm.last_valid_H, om.last_valid_H)
        assert len(vm.homography_history) == len(om.history)
        assert all(np.array_equal(x, y) for x, y in zip(vm.homography_history, om.history))
        assert np.array_equal(vm.output_img_u8(), om.canvas), "frame %d" % t
    vm.close()


@pytest.mark.gpu
@pytest.mark.parametrize("batch", [1, 4, 7])
def test_rejected_frames_through_the_batched_stream_match_oracle(batch):
    """rtvm_process_batch / k_batch_glue: status bit RTVM_FRAME_INVALID_H, the identity in the smoothing
    history (visible through the homographies of the following frames) and the canvas, against the oracle."""
    import torch
    frames, labels = rc.sequence()
    ctx = rtvm.Context(rc.FH, rc.FW, 2 * rc.FH, int(1.2 * rc.FW), nfeatures=500)
    om = OracleMosaic(frames[0], nfeatures=500, glue="explicit")
    dev = [torch.from_numpy(f).cuda() for f in frames]
    torch.cuda.synchronize()
    ctx.stream_begin(dev[0].data_ptr(), om.w_offset, om.h_offset)
    want = []
    for f in frames[1:]:
        om.process_frame(f)
        want.append(om.status)
    assert [bool(s & INVALID) for s in want] != [False] * len(want)
    for lab, s in zip(labels, want):
        assert lab is None or s == INVALID
    if batch == 1:
        got = []
        for d in dev[1:]:
            ctx.process_frame(d.data_ptr())
            got.append(ctx.frame_status()[0])
    else:
        got = ctx.process_video([d.data_ptr() for d in dev[1:]], batch=batch).tolist()
    assert got == want
    st, nk, nm, H = ctx.frame_status()
    assert np.array_equal(H, om.H)
    assert np.array_equal(ctx.canvas_read(), om.canvas)
    ctx.close()
