"""HBM traffic per kernel and frame from two rocprofv3 --pmc runs (FETCH_SIZE, WRITE_SIZE) of tools/pmc_run.py.

usage: python tools/pmc_traffic.py OUT.json FRAMES CONFIG DIR_FETCH DIR_WRITE [CALIB_FETCH_DIR CALIB_WRITE_DIR]
Per kernel: the counters summed over all its dispatches (KB, as rocprofv3 reports them), divided by the frames
the run stitched.  hbm_bytes_per_frame = read_factor * FETCH_SIZE + WRITE_SIZE, read_factor = 2: on gfx950
FETCH_SIZE tallies 128-byte read requests at 64 bytes.  With the calibration directories (tools/pmc_calib under
the same two passes) the measured factors for 2, 4, 8 and 16 bytes per lane are recorded too."""
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def summary(dirs):
    tmp = "/tmp/pmc_summary_%d.json" % os.getpid()
    subprocess.check_call([sys.executable, os.path.join(HERE, "pmc_summary.py"), tmp] + list(dirs))
    with open(tmp) as fh:
        s = json.load(fh)
    os.remove(tmp)
    return s


def main():
    out, frames, config, d_fetch, d_write = sys.argv[1], int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5]
    s = summary([d_fetch, d_write])
    kernels = {}
    for k, e in s.items():
        f, w = e.get("FETCH_SIZE"), e.get("WRITE_SIZE")
        if f is None or w is None or not k.startswith("k_"):
            continue
        kernels[k] = {"dispatches": e.get("dispatches"), "fetch_size_kb": f, "write_size_kb": w,
                      "hbm_bytes_per_frame": (2 * f + w) * 1024.0 / frames}
    calib = None
    if len(sys.argv) > 7:
        c = summary([sys.argv[6], sys.argv[7]])
        gib = float(1 << 30)
        calib = {}
        for k, e in c.items():
            if k.startswith("k_calib_read") and "FETCH_SIZE" in e:
                calib[k] = {"bytes_read": gib, "FETCH_SIZE_bytes": e["FETCH_SIZE"] * 1024.0,
                            "true_over_reported": gib / (e["FETCH_SIZE"] * 1024.0)}
            if k.startswith("k_calib_write") and "WRITE_SIZE" in e:
                calib[k] = {"bytes_written": gib, "WRITE_SIZE_bytes": e["WRITE_SIZE"] * 1024.0,
                            "true_over_reported": gib / (e["WRITE_SIZE"] * 1024.0)}
    about = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, two separate runs of `python3 tools/pmc_run.py %d %s`: "
             "per kernel the counters summed over all dispatches, divided by the %d frames every kernel served.  "
             "hbm_bytes_per_frame = 2 * FETCH_SIZE + WRITE_SIZE (FETCH_SIZE on gfx950 tallies 128-byte read requests at "
             "64 bytes); Infinity-Cache hits are counted.  `calibration`: the same two passes over tools/pmc_calib, "
             "kernels that stream 1 GiB once with 2, 4, 8 and 16 bytes per lane." % (frames, config, frames))
    with open(out, "w") as fh:
        json.dump({"_about": about, "config": config, "frames": frames, "calibration": calib, "kernels": kernels}, fh, indent=1)


if __name__ == "__main__":
    main()
