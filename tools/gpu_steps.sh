#!/bin/bash
# usage: tools/gpu_steps.sh <logdir> <name> <seconds> <command...>  -- runs one GPU step under a time limit with
# its output in gpurun_out/<logdir>/<name>.log, prints the tail, and fails when the step fails (chain calls with &&).
dir=gpurun_out/$1; name=$2; limit=$3; shift 3
mkdir -p "$dir"
timeout -k 10 "$limit" "$@" > "$dir/$name.log" 2>&1
rc=$?
tail -n "${TAIL:-15}" "$dir/$name.log"
echo "[step $name] exit $rc"
exit $rc
