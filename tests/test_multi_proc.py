"""Several stitching processes on one GPU, as `bench.py --gpus N` / `--procs N` start them (one process, one context,
one video each): what each of them stitches while the others run equals what it stitches alone."""
import json
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _start(seed, frames, *shape):
    return subprocess.Popen([sys.executable, os.path.join(HERE, "multi_proc_helper.py"), str(seed), str(frames)] +
                            [str(v) for v in shape],
                            stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)


def _result(p):
    out, err = p.communicate(timeout=300)
    assert p.returncode == 0, err[-2000:]
    return json.loads([l for l in out.splitlines() if l.startswith("{")][-1])


@pytest.mark.gpu
@pytest.mark.parametrize("n,shape", [(96, ()), (40, (1080, 1920, 1000))])
def test_two_processes_side_by_side_stitch_what_each_stitches_alone(n, shape):
    """360 x 640 / 500 features, and the per-GPU shape of the eight-stream configuration (1080p, 1000 features)."""
    alone = [_result(_start(seed, n, *shape)) for seed in (3, 4)]
    procs = [_start(seed, n, *shape) for seed in (3, 4)]       # both at once on the one GPU
    together = [_result(p) for p in procs]
    for a, t in zip(alone, together):
        assert t["status"] == a["status"] == [0] * n
        assert t["matches"] == a["matches"] and t["matches"] > 100
        assert t["H"] == a["H"]
        assert t["canvas_sha1"] == a["canvas_sha1"]
    assert alone[0]["canvas_sha1"] != alone[1]["canvas_sha1"]   # two different videos
