"""The CPU restatement against its own committed checksums (tests/golden/oracle_stage_checksums.json, written by
tests/golden/make_golden.py): every stage of the path on small seeded inputs.  The reference ships no fixtures
(parity with OpenCV itself is unpinned); these pin the restatement, which the GPU parity tests follow."""
import hashlib
import json
import os
import sys

import numpy as np
import pytest

import oracle_ffi as orc

GOLD = os.path.join(os.path.dirname(__file__), "golden")
sys.path.insert(0, GOLD)
import make_golden  # noqa: E402


@pytest.mark.parametrize("threads", [1, 5])
def test_oracle_stages_match_the_committed_checksums(threads):
    """With one thread (the sequential restatement) and with several (loops over independent pixels, rows and
    descriptors split by OpenMP): the same bits."""
    before = orc.set_threads(threads)
    try:
        _check_against_the_checksums()
    finally:
        orc.set_threads(before)


def _check_against_the_checksums():
    with open(os.path.join(GOLD, "oracle_stage_checksums.json")) as fh:
        want = json.load(fh)
    got = make_golden.stage_outputs(orc, make_golden.stage_inputs())
    assert sorted(got) == sorted(want)
    for k, v in got.items():
        v = np.ascontiguousarray(v)
        assert list(v.shape) == want[k]["shape"] and str(v.dtype) == want[k]["dtype"], k
        assert hashlib.sha256(v.tobytes()).hexdigest() == want[k]["sha256"], k
    # the inputs exercise the stages: features found, outliers rejected, a real overlap blended
    assert want["orb_xy"]["shape"][0] > 100 and want["sift_xy"]["shape"][0] > 50
    assert 0 < int(np.asarray(got["ransac_mask"]).sum()) < 200
