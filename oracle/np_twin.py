"""numpy twins of parts of the CPU restatement.  Test infrastructure only.

Written independently of oracle/*.cpp (vectorised brute force instead of
loops) so that the two can check each other and generate the known-answer
fixtures under tests/golden/.  Parity unpinned (see oracle/oracle.h).
"""
import numpy as np

_POP = np.array([bin(i).count("1") for i in range(256)], np.int32)


def hamming_matrix(q, t):
    q = np.asarray(q, np.uint8).reshape(-1, 32)
    t = np.asarray(t, np.uint8).reshape(-1, 32)
    return _POP[q[:, None, :] ^ t[None, :, :]].sum(axis=2).astype(np.int32)


def match_hamming_xcheck(q, t, sorted_by_distance=False):
    """main.py:37,695(,698): mutual nearest neighbours, first minimum on ties."""
    q = np.asarray(q, np.uint8).reshape(-1, 32)
    t = np.asarray(t, np.uint8).reshape(-1, 32)
    if len(q) == 0 or len(t) == 0:
        z = np.zeros(0, np.int32)
        return z, z, z
    D = hamming_matrix(q, t)
    s = D.argmin(axis=1)           # numpy argmin returns the first minimum
    r = D.argmin(axis=0)
    qi = np.nonzero(r[s] == np.arange(len(q)))[0].astype(np.int32)
    ti = s[qi].astype(np.int32)
    di = D[qi, ti].astype(np.int32)
    if sorted_by_distance:
        o = np.argsort(di, kind="stable")
        qi, ti, di = qi[o], ti[o], di[o]
    return qi, ti, di


def match_l2_knn2(q, t):
    """main.py:34,688: two nearest by float32 L2, earlier index first on ties."""
    q = np.asarray(q, np.float32)
    t = np.asarray(t, np.float32)
    diff = q[:, None, :].astype(np.float32) - t[None, :, :]
    d2 = np.zeros(diff.shape[:2], np.float32)
    for k in range(diff.shape[2]):          # sequential float32 accumulation
        d2 = (d2 + diff[:, :, k] * diff[:, :, k]).astype(np.float32)
    d = np.sqrt(d2).astype(np.float32)
    o = np.argsort(d, axis=1, kind="stable")[:, :2]
    return o.astype(np.int32), np.take_along_axis(d, o, axis=1)
