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


def stage_inputs():
    """Small seeded inputs of every stage of the path (shared by the generator and tests/test_golden_stages.py)."""
    rng = np.random.default_rng(2024)
    yy, xx = np.mgrid[0:120, 0:160]
    img = (96 + 60 * np.sin(xx / 7.0) * np.cos(yy / 9.0)).astype(np.float64)
    for _ in range(60):
        x, y, w, h = rng.integers(0, 140), rng.integers(0, 100), rng.integers(4, 24), rng.integers(4, 24)
        img[y:y + h, x:x + w] = rng.integers(1, 256)
    frame = np.clip(np.stack([img, np.roll(img, 3, 1), np.roll(img, 5, 0)], -1), 1, 255).astype(np.uint8)
    H = np.array([[1.01, 0.02, 11.5], [-0.015, 0.99, 7.25], [1e-5, -2e-5, 1.0]])
    src = (rng.random((200, 2)) * [160, 120]).astype(np.float32)
    p = np.c_[src, np.ones(200)] @ H.T
    dst = (p[:, :2] / p[:, 2:]).astype(np.float32)
    dst[::7] += rng.normal(0, 25, dst[::7].shape).astype(np.float32)   # outliers
    mask = (rng.random((70, 90)) > 0.02).astype(np.uint8) * 255
    wts = rng.random((50, 64)).astype(np.float32)
    return dict(frame=frame, H=H, src=src, dst=dst, mask=mask, wts=wts)


def stage_outputs(orc, inp):
    """Every oracle stage on the inputs above: {name: array}."""
    f = inp["frame"]
    out = {"gray": orc.gray(f)}
    o = orc.orb(f, 300)
    out.update({"orb_xy": o["xy"], "orb_angle": o["angle"], "orb_response": o["response"], "orb_desc": o["desc"]})
    s = orc.sift(f, 200)
    out.update({"sift_xy": s["xy"], "sift_size": s["size"], "sift_angle": s["angle"], "sift_desc": s["desc"]})
    Hh, m = orc.find_homography(inp["src"], inp["dst"], 2.0)
    out.update({"ransac_H": Hh, "ransac_mask": m})
    out["warp_float"] = orc.warp_perspective(f, inp["H"], 200, 220, 0)
    out["warp_fixed"] = orc.warp_perspective(f, inp["H"], 200, 220, 1)
    out["dt"] = orc.distance_transform(inp["mask"])
    out["blur31"] = orc.gaussian_blur31(inp["wts"])
    canvas = np.zeros((200, 220, 3), np.uint8)
    canvas[40:160, 30:190] = f
    out["blend"] = orc.warp_blend(canvas, f[::-1].copy(), inp["H"], 0)
    out["resize"] = orc.resize_linear(f, 233, 77)
    out["crop_rect"] = np.array(orc.crop_rect(out["blend"], 15), np.int32)
    return out


def golden_stages():
    """SHA-256 of every oracle stage's output on the seeded inputs: pins the CPU restatement to itself, so that a
    change of its arithmetic (which the GPU parity tests would simply follow) does not go unnoticed."""
    import hashlib
    import json
    import oracle_ffi as orc
    out = stage_outputs(orc, stage_inputs())
    sums = {k: {"shape": list(np.shape(v)), "dtype": str(np.asarray(v).dtype),
                "sha256": hashlib.sha256(np.ascontiguousarray(v).tobytes()).hexdigest()} for k, v in out.items()}
    with open(os.path.join(HERE, "oracle_stage_checksums.json"), "w") as fh:
        json.dump(sums, fh, indent=1, sort_keys=True)


if __name__ == "__main__":
    golden_match()
    golden_stages()
    print("golden fixtures written to", HERE)
