#!/bin/bash
# Runs on the GPU box: the bench lines and rocprofv3 kernel statistics this round's profiles/ holds.
# ONLY=<regex> runs the steps whose name matches.  Results under gpurun_out/$ROUND/ (default r3); every step under its own time limit, the script stops at the first failure.
set -u
ROOT=$(pwd)
R=${ROUND:-r3}
OUT=$ROOT/gpurun_out/$R
mkdir -p "$OUT"
export TMPDIR=/tmp
step() {  # name, limit, command...
    local name=$1 limit=$2; shift 2
    if [ -n "${ONLY:-}" ] && ! [[ $name =~ $ONLY ]]; then return 0; fi
    timeout -k 10 "$limit" "$@" > "$OUT/$name.log" 2>&1
    local rc=$?
    echo "[$name] exit $rc"; tail -n "${TAIL:-2}" "$OUT/$name.log" | cut -c1-1500
    [ $rc -eq 0 ] || exit $rc
}
cd "$ROOT"
step bench_720p_orb1000 400 python3 bench.py --kernel-table "$OUT/kernels_720p_orb1000.json"
step bench_720p_orb1000_driver_command 300 python3 bench.py --gpus 1 --steps 20 --warmup 5
step bench_720p_sift700 300 python3 bench.py --config 720p_sift700 --cpu-frames 4 --kernel-table "$OUT/kernels_720p_sift700.json"
step bench_1080p_sift2000_l2 300 python3 bench.py --config 1080p_sift2000_l2 --cpu-frames 4
step bench_4k_orb4000_canvas16k 400 python3 bench.py --config 4k_orb4000_canvas16k --steps 64 --warmup 8 --cpu-frames 1 --host-frames 0 --dropin-frames 0 --sift-frames 0 --kernel-table "$OUT/kernels_4k_orb4000_canvas16k.json"
step bench_1080p_orb1000 300 python3 bench.py --config 1080p_orb1000 --steps 256 --cpu-frames 4 --kernel-table "$OUT/kernels_1080p_orb1000.json"
step bench_360p_orb700 300 python3 bench.py --config 360p_orb700 --steps 256 --cpu-frames 8 --kernel-table "$OUT/kernels_360p_orb700.json"
step bench_procs2 300 python3 bench.py --procs 2 --cpu-frames 0 --host-frames 0 --dropin-frames 0 --sift-frames 0
step bench_procs4 300 python3 bench.py --procs 4 --cpu-frames 0 --host-frames 0 --dropin-frames 0 --sift-frames 0
cd /tmp
step stats_720p_orb1000 300 rocprofv3 --kernel-trace --stats -d "$OUT/stats_720p_orb1000" --output-format csv -- python3 "$ROOT/bench.py" --steps 256 --cpu-frames 0 --host-frames 0 --dropin-frames 0 --sift-frames 0
step stats_720p_orb1000_driver_command 300 rocprofv3 --kernel-trace --stats -d "$OUT/stats_720p_orb1000_driver_command" --output-format csv -- python3 "$ROOT/bench.py" --gpus 1 --steps 20 --warmup 5
step stats_720p_sift700 300 rocprofv3 --kernel-trace --stats -d "$OUT/stats_720p_sift700" --output-format csv -- python3 "$ROOT/bench.py" --config 720p_sift700 --steps 128 --cpu-frames 0
step stats_1080p_sift2000_l2 300 rocprofv3 --kernel-trace --stats -d "$OUT/stats_1080p_sift2000_l2" --output-format csv -- python3 "$ROOT/bench.py" --config 1080p_sift2000_l2 --cpu-frames 0
step stats_4k_orb4000_canvas16k 300 rocprofv3 --kernel-trace --stats -d "$OUT/stats_4k_orb4000_canvas16k" --output-format csv -- python3 "$ROOT/bench.py" --config 4k_orb4000_canvas16k --steps 32 --warmup 8 --cpu-frames 0 --host-frames 0 --dropin-frames 0 --sift-frames 0
echo "profiles done"
