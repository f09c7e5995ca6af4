#!/bin/bash
# Runs on the GPU box: the rocprofv3 counter passes (one counter per run) over the calibration program and over the
# 48-frame pipeline workload, then the kernel trace of the default bench.  Results under gpurun_out/$ROUND/ (default r3).
set -u
ROOT=$(pwd)
R=${ROUND:-r3}
OUT=$ROOT/gpurun_out/$R
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
step() {  # name, limit, command...
    local name=$1 limit=$2; shift 2
    timeout -s KILL "$limit" "$@" > "$OUT/$name.log" 2>&1
    local rc=$?
    echo "[$name] exit $rc"; tail -n 2 "$OUT/$name.log"
    [ $rc -eq 0 ] || exit $rc
}
CFG=${CONFIG:-720p_orb1000}
if [ "$CFG" = 720p_orb1000 ]; then
step pmc_calib_fetch 120 rocprofv3 --pmc FETCH_SIZE -d "$OUT/pmc_calib_fetch" --output-format csv -- "$ROOT/tools/bin/pmc_calib"
step pmc_calib_write 120 rocprofv3 --pmc WRITE_SIZE -d "$OUT/pmc_calib_write" --output-format csv -- "$ROOT/tools/bin/pmc_calib"
fi
rm -rf "$OUT/pmc_fetch_$CFG" "$OUT/pmc_write_$CFG"
step pmc_fetch_$CFG 240 rocprofv3 --pmc FETCH_SIZE -d "$OUT/pmc_fetch_$CFG" --output-format csv -- python3 "$ROOT/tools/pmc_run.py" 48 $CFG
step pmc_write_$CFG 240 rocprofv3 --pmc WRITE_SIZE -d "$OUT/pmc_write_$CFG" --output-format csv -- python3 "$ROOT/tools/pmc_run.py" 48 $CFG
cd "$ROOT"
if [ "$CFG" = 720p_orb1000 ]; then
python3 tools/pmc_traffic.py "$OUT/${R}_pmc_traffic.json" 48 $CFG "$OUT/pmc_fetch_$CFG" "$OUT/pmc_write_$CFG" "$OUT/pmc_calib_fetch" "$OUT/pmc_calib_write" || exit 1
else
python3 tools/pmc_traffic.py "$OUT/${R}_pmc_traffic_$CFG.json" 48 $CFG "$OUT/pmc_fetch_$CFG" "$OUT/pmc_write_$CFG" || exit 1
fi
echo "traffic summary written"
