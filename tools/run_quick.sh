#!/bin/bash
# Runs on the GPU box: the tests named in $TESTS, then the default bench (all streams) and the same bench with every
# kernel on one stream (kernel times without other kernels beside them), with kernel tables under gpurun_out/r2/.
set -u
T=${TESTS:-tests/test_dt.py tests/test_warp_blend.py tests/test_mosaic.py}
S=${SUFFIX:-q}
TAIL=6 tools/gpu_steps.sh r2 t_$S 600 python -u -m pytest $T -x -q -m gpu --timeout 170 --timeout-method thread || exit 1
ARGS="--cpu-frames 0 --host-frames 0 --dropin-frames 0 --sift-frames 0"
TAIL=1 tools/gpu_steps.sh r2 b_$S 300 python bench.py $ARGS --kernel-table gpurun_out/r2/k_$S.json ${BENCH_ARGS:-} || exit 1
TAIL=1 tools/gpu_steps.sh r2 b1_$S 300 python bench.py $ARGS --single-stream --steps 256 --kernel-table gpurun_out/r2/k1_$S.json ${BENCH_ARGS:-} || exit 1
