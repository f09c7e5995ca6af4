"""bench.py under torch.distributed.run with two ranks on CPU (gloo): the launch contract of the multi-GPU run --
RANK / LOCAL_RANK / WORLD_SIZE from the environment, the process group, the barriers around the timed region, the
max over ranks, the per-rank rates, one JSON line from rank 0 -- with a stand-in context (tests/bench_stub.py), since
this box has no GPU, let alone two.  What the ranks would stitch is covered by the GPU tests; this covers the rest."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_bench_two_ranks_under_torch_distributed_run():
    env = dict(os.environ, RTVM_BENCH_STUB="1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "40", "--warmup", "8"]
    p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout          # rank 0 alone prints, one line
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 40 and out["warmup"] == 8 and out["scaling"] == "weak"
    assert out["unit"] == "frames/s" and out["higher_is_better"] is True and out["vs_baseline"] is None
    assert out["config"]["streams"] == 2 and out["config"]["parallelism"] == "one stream per GPU"
    rates = out["per_process_frames_per_s"]
    assert len(rates) == 2 and 600 < rates[0] < 1100 and 300 < rates[1] < 550      # the stand-in's 1000 and 500 frames/s
    # whole-job value: all ranks' frames over the slowest rank's time (plus the closing barrier)
    assert 0.75 * 2 * rates[1] <= out["value"] <= 2 * rates[1] * 1.02
    assert abs(out["ms_per_step"] - 1e3 * 2 / out["value"]) < 1e-6
    assert "stand-in" in out["data"] and out["cpu_baseline"] is None      # never mistaken for a measurement
