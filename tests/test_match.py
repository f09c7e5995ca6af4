"""Matcher: oracle vs numpy twin vs golden fixtures (CPU), HIP vs oracle (GPU)."""
import os
import sys

import numpy as np
import pytest

import oracle_ffi as orc
sys.path.insert(0, os.path.join(orc.ROOT, "oracle"))
import np_twin  # noqa: E402

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def make_desc(seed, nq, nt, planted=0.6, noise_bits=20, ties=True):
    """Random 256-bit descriptors with planted correspondences and exact duplicates."""
    rng = np.random.default_rng(seed)
    t = rng.integers(0, 256, (nt, 32), dtype=np.uint8)
    q = rng.integers(0, 256, (nq, 32), dtype=np.uint8)
    k = int(min(nq, nt) * planted)
    src = rng.permutation(nt)[:k]
    dst = rng.permutation(nq)[:k]
    q[dst] = t[src]
    for i in dst:  # flip a few bits
        nb = rng.integers(0, noise_bits + 1)
        bits = rng.integers(0, 256, nb)
        for b in bits:
            q[i, b >> 3] ^= np.uint8(1 << (b & 7))
    if ties and nt >= 8 and nq >= 8:
        t[3] = t[1]          # duplicate train rows: first index must win
        t[7] = t[1]
        q[5] = q[2]          # duplicate query rows
        q[0] = t[1]
    return q, t


CASES = [(0, 0), (0, 5), (5, 0), (1, 1), (3, 70), (70, 3), (64, 64), (65, 63), (700, 700), (1000, 900)]


@pytest.mark.parametrize("nq,nt", CASES)
@pytest.mark.parametrize("srt", [False, True])
def test_oracle_matches_numpy_twin(nq, nt, srt):
    q, t = make_desc(nq * 131 + nt, nq, nt)
    a = orc.match_hamming(q, t, srt)
    b = np_twin.match_hamming_xcheck(q, t, srt)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)


def test_oracle_golden_fixture():
    g = np.load(os.path.join(GOLD, "match_hamming.npz"))
    for tag in ("a", "b"):
        got = orc.match_hamming(g["q_" + tag], g["t_" + tag], True)
        assert np.array_equal(got[0], g["qi_" + tag])
        assert np.array_equal(got[1], g["ti_" + tag])
        assert np.array_equal(got[2], g["di_" + tag])


def test_mutual_nn_property():
    q, t = make_desc(7, 300, 280)
    qi, ti, di = orc.match_hamming(q, t)
    D = np_twin.hamming_matrix(q, t)
    assert np.all(np.diff(qi) > 0)
    for i, j, d in zip(qi, ti, di):
        assert D[i, j] == d == D[i].min() == D[:, j].min()
    assert len(set(ti.tolist())) == len(ti)


def test_l2_oracle_matches_numpy_twin():
    rng = np.random.default_rng(3)
    t = rng.integers(0, 256, (150, 128)).astype(np.float32)
    q = rng.integers(0, 256, (120, 128)).astype(np.float32)
    q[:60] = t[:60] + rng.integers(-3, 4, (60, 128))
    t[9] = t[4]
    q[1] = t[4]
    a = orc.match_l2_knn2(q, t)
    b = np_twin.match_l2_knn2(q, t)
    assert np.array_equal(a[0], b[0])
    assert np.array_equal(a[1], b[1])


# ---------------------------------------------------------------- GPU parity
@pytest.mark.gpu
@pytest.mark.parametrize("nq,nt", CASES + [(4000, 4000), (4097, 3999)])
@pytest.mark.parametrize("srt", [False, True])
def test_hip_hamming_bit_exact(ctx_small, nq, nt, srt):
    q, t = make_desc(nq * 17 + nt + 1, nq, nt)
    got = ctx_small.match_hamming(q, t, srt)
    ref = orc.match_hamming(q, t, srt)
    for x, y in zip(got, ref):
        assert np.array_equal(x, y)


@pytest.mark.gpu
def test_hip_hamming_golden(ctx_small):
    g = np.load(os.path.join(GOLD, "match_hamming.npz"))
    for tag in ("a", "b"):
        got = ctx_small.match_hamming(g["q_" + tag], g["t_" + tag], True)
        assert np.array_equal(got[0], g["qi_" + tag])
        assert np.array_equal(got[1], g["ti_" + tag])
        assert np.array_equal(got[2], g["di_" + tag])


@pytest.mark.gpu
def test_hip_hamming_all_identical(ctx_small):
    """Every distance ties at 0: lowest index wins in both directions -> one match (0,0)."""
    q = np.zeros((100, 32), np.uint8)
    t = np.zeros((90, 32), np.uint8)
    qi, ti, di = ctx_small.match_hamming(q, t, True)
    assert qi.tolist() == [0] and ti.tolist() == [0] and di.tolist() == [0]


@pytest.mark.gpu
def test_hip_hamming_rejects_oversize(ctx_small):
    import rtvm
    q = np.zeros((16385, 32), np.uint8)
    with pytest.raises(rtvm.RtvmError):
        ctx_small.match_hamming(q, q[:10])


def sift_like(seed, nq, nt, dim=128, planted=0.7):
    """Integer-valued float32 descriptors (0..255) with planted correspondences, as SIFT produces."""
    rng = np.random.default_rng(seed)
    t = rng.integers(0, 120, (nt, dim)).astype(np.float32)
    q = rng.integers(0, 120, (nq, dim)).astype(np.float32)
    k = int(min(nq, nt) * planted)
    src, dst = rng.permutation(nt)[:k], rng.permutation(nq)[:k]
    q[dst] = np.clip(t[src] + np.rint(rng.normal(0, 8, (k, dim))), 0, 255).astype(np.float32)
    if nt > 12 and nq > 3:
        t[9] = t[4]           # exact duplicates: the earlier index must come first
        t[11] = t[4]
        q[1] = t[4]
    return q, t


@pytest.mark.gpu
@pytest.mark.parametrize("nq,nt,dim", [(1, 2, 128), (33, 31, 128), (500, 700, 128), (2000, 2000, 128),
                                       (300, 257, 64), (100, 90, 32), (128, 128, 128), (129, 257, 128),
                                       (4100, 3000, 128)])
def test_hip_l2_knn2_integer_valued_exact(ctx_small, nq, nt, dim):
    q, t = sift_like(nq + nt, nq, nt, dim)
    gi, gd = ctx_small.match_l2_knn2(q, t)
    ri, rd = orc.match_l2_knn2(q, t)
    assert np.array_equal(gi, ri)
    assert np.array_equal(gd, rd)


@pytest.mark.gpu
def test_hip_l2_knn2_far_descriptors_order_by_the_rounded_distance(ctx_small):
    """From d2 = 2^22 up neighbouring integers can round to the same float32 root; OpenCV compares the roots, so
    the earlier train row wins a tie of roots although its d2 is larger.  Mixed with queries whose neighbours are
    near (the common case, ordered by d2 on the matrix cores)."""
    rng = np.random.default_rng(17)
    nt, nq = 300, 140
    t = np.zeros((nt, 128), np.float32)
    t[:, :100] = 255                                   # 100 * 255^2 = 6 502 500 > 2^22 from an all-zero query
    t[:, 100:] = rng.integers(0, 2, (nt, 28))          # plus 0..28: roots 0.8 ulp apart
    t[:, 100:108] = 1                                  # at least 8 everywhere ...
    t[5, 100:] = 0
    t[5, 100:103] = 1                                  # ... except row 5: d2 = 6 502 503 ...
    t[9, 100:] = 0
    t[9, 100:102] = 1                                  # ... and row 9: 6 502 502, the same float32 root
    q = np.zeros((nq, 128), np.float32)
    q[40:, :100] = 255                                 # near queries: d2 = 0..28 + noise
    q[40:, 100:] = rng.integers(0, 3, (nq - 40, 28))
    q[20:40, :64] = 255                                # in between: 36 * 255^2 = 2 340 900 < 2^22
    ri, rd = orc.match_l2_knn2(q, t)
    d2 = ((q[0][None] - t) ** 2).sum(1)
    assert int(np.argmin(d2)) == 9 and ri[0].tolist() == [5, 9] and rd[0, 0] == rd[0, 1]   # the earlier row wins
    gi, gd = ctx_small.match_l2_knn2(q, t)
    assert np.array_equal(gi, ri)
    assert np.array_equal(gd, rd)


@pytest.mark.gpu
def test_hip_l2_knn2_generic_floats_exact(ctx_small):
    rng = np.random.default_rng(5)
    t = rng.normal(0, 1, (400, 128)).astype(np.float32)
    q = (t[rng.integers(0, 400, 350)] + rng.normal(0, 0.05, (350, 128))).astype(np.float32)
    gi, gd = ctx_small.match_l2_knn2(q, t)
    ri, rd = orc.match_l2_knn2(q, t)
    assert np.array_equal(gi, ri) and np.array_equal(gd, rd)
    q2 = q.copy()
    q2[0, 0] = 300.0                      # integer valued but out of the 8-bit range
    gi, gd = ctx_small.match_l2_knn2(np.rint(q2 * 50), np.rint(t * 50))
    ri, rd = orc.match_l2_knn2(np.rint(q2 * 50), np.rint(t * 50))
    assert np.array_equal(gi, ri) and np.array_equal(gd, rd)


@pytest.mark.gpu
def test_hip_l2_knn2_needs_two_train_rows(ctx_small):
    import rtvm
    q, t = sift_like(1, 5, 5)
    with pytest.raises(rtvm.RtvmError):
        ctx_small.match_l2_knn2(q, t[:1])


def test_legacy_cross_check_oracle_semantics():
    """Before OpenCV 4.2 a query was matched with the nearest of the train rows that chose it, whether or not that row
    was the query's own nearest: a superset of the queries of the mutual rule, with the same pairs where both apply."""
    rng = np.random.default_rng(23)
    t = rng.integers(0, 256, (300, 32), dtype=np.uint8)
    q = t[rng.integers(0, 300, 260)].copy()
    q ^= (rng.random(q.shape) < 0.08).astype(np.uint8) * np.uint8(1 << 3)
    mq, mt, md = orc.match_hamming(q, t)
    lq, lt, ld = orc.match_hamming_legacy(q, t)
    assert set(mq.tolist()) < set(lq.tolist())            # strictly more queries matched
    mutual = dict(zip(mq.tolist(), zip(mt.tolist(), md.tolist())))
    legacy = dict(zip(lq.tolist(), zip(lt.tolist(), ld.tolist())))
    assert all(legacy[i] == mutual[i] for i in mutual)
    # brute force: per train row its nearest query (first minimum), per query the nearest of those rows
    D = np.unpackbits(q[:, None, :] ^ t[None, :, :], axis=2).sum(2)
    owner = D.argmin(axis=0)
    for i in range(len(q)):
        js = np.nonzero(owner == i)[0]
        if len(js) == 0:
            assert i not in legacy
        else:
            j = js[np.argmin(D[i, js])]
            assert legacy[i] == (int(j), int(D[i, j]))


@pytest.mark.gpu
@pytest.mark.parametrize("nq,nt,seed", [(260, 300, 23), (1000, 1000, 3), (65, 4097, 4), (700, 1, 5), (0, 40, 6)])
def test_hip_hamming_legacy_cross_check_bit_exact(nq, nt, seed):
    import rtvm
    rng = np.random.default_rng(seed)
    t = rng.integers(0, 256, (nt, 32), dtype=np.uint8)
    q = t[rng.integers(0, nt, nq)].copy() if nq else np.zeros((0, 32), np.uint8)
    q ^= (rng.random(q.shape) < 0.08).astype(np.uint8) * np.uint8(1 << 3)
    ctx = rtvm.Context(flags=rtvm._ffi.FLAG_MATCH_LEGACY)
    for srt in (False, True):
        got = ctx.match_hamming(q, t, sorted_by_distance=srt)
        want = orc.match_hamming_legacy(q, t, srt)
        for a, b in zip(got, want):
            assert np.array_equal(a, b)
    ctx.close()
