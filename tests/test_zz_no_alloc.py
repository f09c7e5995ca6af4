"""rtvm_match_hamming_xcheck_noalloc (the entry point for callers that cannot allocate) against
rtvm_match_hamming_xcheck[_sorted], through the C ABI, on two calls in a row (the second with a
smaller input: the fixed scratch keeps the first call's contents)."""
import numpy as np
import pytest

import rtvm

F = rtvm._ffi


def descriptors(rng, n, base=None, flips=20):
    """n random 256-bit descriptors; with `base`, noisy copies of its rows (so that mutual nearest
    neighbours exist) mixed with random ones."""
    d = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    if base is not None and len(base):
        k = min(n, len(base)) * 2 // 3
        rows = rng.permutation(len(base))[:k]
        noisy = base[rows].copy()
        for r in range(k):
            for b in rng.integers(0, 256, flips):
                noisy[r, b // 8] ^= np.uint8(1 << (b % 8))
        d[rng.permutation(n)[:k]] = noisy
    return d


def run_noalloc(q, t, sorted_by_distance, stream=0):
    import torch
    dev = torch.device("cuda", 0)
    nq, nt = len(q), len(t)
    dq = torch.from_numpy(q).to(dev) if nq else torch.zeros(1, dtype=torch.uint8, device=dev)
    dt = torch.from_numpy(t).to(dev) if nt else torch.zeros(1, dtype=torch.uint8, device=dev)
    out = torch.full((3, max(nq, 1)), -7, dtype=torch.int32, device=dev)
    m = torch.full((1,), -7, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    F.match_hamming_noalloc(dq.data_ptr(), nq, dt.data_ptr(), nt, out[0].data_ptr(), out[1].data_ptr(),
                            out[2].data_ptr(), m.data_ptr(), sorted_by_distance, stream)
    torch.cuda.synchronize()
    k = int(m.item())
    o = out.cpu().numpy()
    assert (o[:, k:] == -7).all()          # nothing written past the matches
    return o[0, :k], o[1, :k], o[2, :k]


@pytest.mark.gpu
def test_noalloc_matcher_equals_the_context_matcher_on_consecutive_calls(ctx_small):
    import torch
    rng = np.random.default_rng(41)
    t1 = descriptors(rng, 1500)
    q1 = descriptors(rng, 1700, t1)
    t2 = descriptors(rng, 333)
    q2 = descriptors(rng, 257, t2, flips=40)
    q2[5] = q2[9]                           # ties: the lower index wins
    side = torch.cuda.Stream()
    for q, t, srt, stream in ((q1, t1, True, 0), (q2, t2, True, 0), (q2, t2, False, side.cuda_stream),
                              (q1[:64], t1[:1], True, 0), (q1[:0], t1, True, 0), (q1[:3], t1[:0], False, 0),
                              (q1, t1, False, 0)):
        want = ctx_small.match_hamming(q, t, sorted_by_distance=srt)
        got = run_noalloc(q, t, srt, stream)
        assert len(want[0]) == len(got[0])
        for a, b in zip(want, got):
            assert np.array_equal(a, b)
    assert len(ctx_small.match_hamming(q1, t1)[0]) > 500 and len(ctx_small.match_hamming(q2, t2)[0]) > 50


@pytest.mark.gpu
def test_noalloc_matcher_at_its_limit(ctx_small):
    """The largest input the fixed scratch takes (both sides at RTVM_NOALLOC_MAX_DESCRIPTORS), then a small one."""
    rng = np.random.default_rng(43)
    n = F.NOALLOC_MAX_DESCRIPTORS
    t1 = descriptors(rng, n)
    q1 = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    rows = rng.permutation(n)[:n // 2]
    q1[rows] = t1[rng.permutation(n)[:n // 2]] ^ rng.integers(0, 2, (n // 2, 32), dtype=np.uint8)
    for q, t in ((q1, t1), (q1[:100], t1[:90])):
        want = ctx_small.match_hamming(q, t, sorted_by_distance=True)
        got = run_noalloc(q, t, True)
        assert len(want[0]) == len(got[0]) > len(q) // 4
        for a, b in zip(want, got):
            assert np.array_equal(a, b)


@pytest.mark.gpu
def test_noalloc_matcher_refuses_what_the_fixed_scratch_cannot_hold():
    import torch
    d = torch.zeros((8, 32), dtype=torch.uint8, device="cuda")
    o = torch.zeros(8, dtype=torch.int32, device="cuda")
    big = F.NOALLOC_MAX_DESCRIPTORS + 1
    for nq, nt in ((big, 8), (8, big)):
        with pytest.raises(F.RtvmError) as e:
            F.match_hamming_noalloc(d.data_ptr(), nq, d.data_ptr(), nt, o.data_ptr(), o.data_ptr(), o.data_ptr(), o.data_ptr())
        assert e.value.code == F.E_ARG and "at most %d" % F.NOALLOC_MAX_DESCRIPTORS in str(e.value)
    with pytest.raises(F.RtvmError):
        F.match_hamming_noalloc(d.data_ptr(), 8, d.data_ptr(), 8, o.data_ptr(), o.data_ptr(), o.data_ptr(), 0)
    with pytest.raises(F.RtvmError) as e:   # descriptors are read as 16-byte words
        F.match_hamming_noalloc(d.data_ptr() + 4, 4, d.data_ptr(), 8, o.data_ptr(), o.data_ptr(), o.data_ptr(), o.data_ptr())
    assert e.value.code == F.E_ARG and "aligned" in str(e.value)
    torch.cuda.synchronize()
