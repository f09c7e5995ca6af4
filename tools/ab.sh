#!/bin/bash
# Runs on the GPU box: the default bench with every library under abtest/ in turn (all streams and one stream),
# the rate and the named kernels' lines of each.  Results under gpurun_out/$1/.  Usage: tools/ab.sh <out> <kernel regex>
set -u
OUT=gpurun_out/${1:-ab}
PAT=${2:-k_warp}
mkdir -p "$OUT"
LIB=real-time-video-mosaic_amd/lib/librtvm.so
cp $LIB "$OUT/keep.so"
# whatever happens, the build under test goes back where the package loads it from
trap 'cp "$OUT/keep.so" $LIB; rm -f "$OUT/keep.so"' EXIT
for so in abtest/*.so; do
    tag=$(basename $so .so)
    cp $so $LIB
    for mode in "" "--single-stream"; do
        timeout -k 10 200 python3 bench.py --cpu-frames 0 --host-frames 0 --dropin-frames 0 --sift-frames 0 $mode ${AB_ARGS:-} \
            --kernel-table "$OUT/k_$tag$mode.json" > "$OUT/b_$tag$mode.log" 2>&1 || { tail -5 "$OUT/b_$tag$mode.log"; exit 1; }
        python3 - "$OUT/b_$tag$mode.log" "$OUT/k_$tag$mode.json" "$tag $mode" "$PAT" <<'P'
import json, sys, re
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
t = json.load(open(sys.argv[2]))["kernels"]
ks = [dict(name=k, **v) for k, v in t.items()] if isinstance(t, dict) else t
print("%-28s value %6d" % (sys.argv[3], d["value"]), {k: round(1e3 * v, 1) for k, v in d["roofline"].get("stream_kernel_ms_per_frame", {}).items()})
for k in ks:
    if re.search(sys.argv[4], k["name"]):
        print("    %-22s %7.2f us/frame %8.1f us/launch x %d" % (k["name"], k["us_per_frame"], k["avg_us"], k["launches"]))
P
    done
done
