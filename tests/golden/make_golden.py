"""Generates the known-answer fixtures in this directory from the numpy twins
(oracle/np_twin.py).  The reference ships no fixtures of its own, so these are
the build's own vectors: small seeded inputs plus the outputs of the
vectorised brute-force twin, against which both the C++ restatement and the
HIP kernels are checked.  Re-run: python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import np_twin  # noqa: E402


def golden_match():
    from test_match import make_desc
    out = {}
    for tag, (seed, nq, nt) in {"a": (11, 96, 80), "b": (12, 33, 130)}.items():
        q, t = make_desc(seed, nq, nt)
        qi, ti, di = np_twin.match_hamming_xcheck(q, t, True)
        out.update({"q_" + tag: q, "t_" + tag: t, "qi_" + tag: qi, "ti_" + tag: ti, "di_" + tag: di})
    np.savez_compressed(os.path.join(HERE, "match_hamming.npz"), **out)


if __name__ == "__main__":
    golden_match()
    print("golden fixtures written to", HERE)
