#!/bin/bash
# Runs on the GPU box: two rocprofv3 counter passes of shader-core counters over the 48-frame pipeline workload
# (kernels run one at a time under the profiler) and a kernel trace of the same workload for the durations.
# Results: gpurun_out/sq/sq.json (per kernel: counters summed over dispatches) and gpurun_out/sq/trace/.
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/${SQ_OUT:-sq}
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
CFG=${CONFIG:-720p_orb1000}
rm -rf "$OUT/p1" "$OUT/p2" "$OUT/trace"
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d "$OUT/p1" --output-format csv -- python3 "$ROOT/tools/pmc_run.py" 48 $CFG > "$OUT/p1.log" 2>&1 || { tail -5 "$OUT/p1.log"; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_ANY -d "$OUT/p2" --output-format csv -- python3 "$ROOT/tools/pmc_run.py" 48 $CFG > "$OUT/p2.log" 2>&1 || { tail -5 "$OUT/p2.log"; exit 1; }
timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d "$OUT/trace" --output-format csv -- python3 "$ROOT/tools/pmc_run.py" 48 $CFG > "$OUT/trace.log" 2>&1 || { tail -5 "$OUT/trace.log"; exit 1; }
cd "$ROOT"
python3 tools/pmc_summary.py "$OUT/sq.json" "$OUT/p1" "$OUT/p2" || exit 1
find "$OUT/trace" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats.csv" \;
find "$OUT" -name "*.csv" -size +2M -delete
echo done
