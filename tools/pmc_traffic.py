"""HBM traffic per kernel from two rocprofv3 --pmc runs (FETCH_SIZE, WRITE_SIZE).

usage: python tools/pmc_traffic.py OUT.json FRAMES_PER_LAUNCH DIR_FETCH DIR_WRITE
Writes, per kernel, the largest per-dispatch FETCH_SIZE and WRITE_SIZE (KB, as rocprofv3 reports them: the
dispatch that serves a full batch) and hbm_bytes = 2 * FETCH_SIZE + WRITE_SIZE in bytes.
"""
import json
import subprocess
import sys
import os

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    out, fpl, d_fetch, d_write = sys.argv[1], int(sys.argv[2]), sys.argv[3], sys.argv[4]
    tmp = out + ".tmp"
    subprocess.check_call([sys.executable, os.path.join(HERE, "pmc_summary.py"), tmp, d_fetch, d_write])
    with open(tmp) as fh:
        s = json.load(fh)
    os.remove(tmp)
    kernels = {}
    for k, e in s.items():
        f, w = e.get("FETCH_SIZE_max"), e.get("WRITE_SIZE_max")
        if f is None or w is None:
            continue
        kernels[k] = {"fetch_size_kb": f, "write_size_kb": w, "hbm_bytes": (2 * f + w) * 1024.0}
    about = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (two separate runs of `python3 bench.py --cpu-frames 0 "
             "--dropin-frames 0 --steps 64 --warmup 32`, 720p_orb1000, batch %d): largest per-dispatch value of each "
             "kernel (a launch that serves a full batch), in KB as rocprofv3 reports them.  hbm_bytes = 2 * FETCH_SIZE "
             "+ WRITE_SIZE: FETCH_SIZE on gfx950 tallies 128-byte read requests at 64 bytes (calibrated for "
             "16-byte-per-lane streaming reads; narrower loads are uncalibrated, so hbm_bytes is an upper estimate of "
             "the read side); Infinity-Cache hits are counted." % fpl)
    with open(out, "w") as fh:
        json.dump({"_about": about, "frames_per_launch": fpl, "kernels": kernels}, fh, indent=1)


if __name__ == "__main__":
    main()
