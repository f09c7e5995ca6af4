"""Prints the per-kernel table bench.py --kernel-table wrote: python tools/show_kernels.py FILE [FILE2] (two files side by side)."""
import json
import sys


def load(p):
    d = json.load(open(p))
    return {k: (v["stream"], v["us_per_frame"]) for k, v in d["kernels"].items()}


a = load(sys.argv[1])
b = load(sys.argv[2]) if len(sys.argv) > 2 else None
names = sorted(set(a) | set(b or {}), key=lambda k: ((a.get(k) or b.get(k))[0], -(a.get(k) or (0, 0))[1]))
tot = {}
for k in names:
    st = (a.get(k) or b.get(k))[0]
    va = a.get(k, (st, 0.0))[1]
    vb = b.get(k, (st, 0.0))[1] if b else None
    t = tot.setdefault(st, [0.0, 0.0])
    t[0] += va
    t[1] += vb or 0.0
    if va < 0.5 and (vb or 0) < 0.5:
        continue
    print("%d %-26s %8.2f" % (st, k, va) + ("  %8.2f" % vb if b else ""))
for st, t in sorted(tot.items()):
    print("stream %d total %8.2f  %8.2f" % (st, t[0], t[1]))
