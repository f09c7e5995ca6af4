#!/bin/bash
# Runs on the GPU box: the GPU test suite, then the default bench with its kernel table.  Results under gpurun_out/$1/.
set -u
OUT=gpurun_out/${1:-check}
mkdir -p "$OUT"
timeout -k 10 ${TEST_LIMIT:-600} python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread ${PYTEST_ARGS:-} > "$OUT/gputest.log" 2>&1
rc=$?
tail -n 4 "$OUT/gputest.log"
[ $rc -eq 0 ] || { grep -n "Error\|error\|assert" "$OUT/gputest.log" | head -20; exit $rc; }
timeout -k 10 300 python3 bench.py ${BENCH_ARGS:-} --kernel-table "$OUT/kernels.json" > "$OUT/bench.log" 2>&1
rc=$?
tail -c 600 "$OUT/bench.log"
[ $rc -eq 0 ] || exit $rc
python3 - "$OUT" <<'P'
import json, sys
out = sys.argv[1]
line = [l for l in open(out + "/bench.log") if l.startswith("{")][-1]
d = json.loads(line)
print("\nvalue", round(d["value"]), "us/frame", round(1e3 * d["ms_per_step"], 2), "parity", d.get("parity_checked"),
      "stage", d.get("warp_blend_stage", {}).get("us_per_frame"))
print({k: round(1e3 * v, 1) for k, v in d["roofline"].get("stream_kernel_ms_per_frame", {}).items()})
t = json.load(open(out + "/kernels.json"))
ks = t["kernels"]
ks = [dict(name=k, **v) for k, v in ks.items()] if isinstance(ks, dict) else ks
ks.sort(key=lambda k: -k["total_ms"])
for k in ks[:26]:
    print("%-22s s%d %7.2f us/frame  %8.1f us/launch x %d" % (k["name"], k["stream"], k["us_per_frame"], k["avg_us"], k["launches"]))
P
if [ -n "${EXTRA_BENCH:-}" ]; then
    timeout -k 10 300 python3 bench.py $EXTRA_BENCH > "$OUT/bench_extra.log" 2>&1 || exit $?
    python3 -c "
import json,sys
d=json.loads([l for l in open('$OUT/bench_extra.log') if l.startswith('{')][-1])
print('extra:', round(d['value']), 'us/frame', round(1e3*d['ms_per_step'],2), 'parity', d.get('parity_checked'))"
fi
