"""Frames that validate_homography (reference main.py:761-801) must reject, one rule at a time.

Every case is a view of the same synthetic world whose homography onto the frame before it breaks
exactly one of the reference's rules: translation above 50 px (main.py:779-789), scale outside
1 +- 0.3 (main.py:784, 792), a perspective term above 1e-3 (main.py:797).  The reference then uses
the identity for that frame (main.py:734-737), still warps it and still advances its state.
Tests only.
"""
import numpy as np

from rtvm import synth

FH, FW = 240, 320
WORLD_H, WORLD_W = 520, 620

# rule index in (translation, scale, perspective) -> views that break only that rule
CASES = {
    "shift": (0, dict(dx=60, dy=-25)),
    "zoom_in": (1, dict(s=0.64)),
    "zoom_out": (1, dict(s=1.4)),
    "persp_x": (2, dict(p=(1.6e-3, 0.0))),
    "persp_y": (2, dict(p=(0.0, -1.5e-3))),
}


def view(dx=0.0, dy=0.0, s=1.0, rot=0.0, p=(0.0, 0.0)):
    """Camera matrix G (frame(p) = world(G p)): zoom, rotation and perspective about the frame's
    origin, then a shift in world coordinates."""
    base = synth.camera_matrix(0, 100.0, WORLD_H - FH - 100.0)
    th = np.deg2rad(rot)
    A = np.array([[np.cos(th) * s, -np.sin(th) * s, 0], [np.sin(th) * s, np.cos(th) * s, 0], [0, 0, 1.0]])
    P = np.array([[1, 0, 0], [0, 1, 0], [p[0], p[1], 1.0]])
    M = np.array([[1, 0, dx], [0, 1, dy], [0, 0, 1.0]]) @ base @ A @ P
    return M / M[2, 2]


def world():
    return synth.make_world(WORLD_H, WORLD_W, seed=0)


def frame(w, **kw):
    return synth.render_frame(w, view(**kw), FH, FW)


def broken_rules(H):
    """Which of the three threshold rules of validate_homography H breaks."""
    with np.errstate(invalid="ignore"):
        scale = np.sqrt(np.linalg.det(H[:2, :2]))
    return (bool(np.hypot(H[0, 2], H[1, 2]) > 50), bool(abs(scale - 1.0) > 0.3),
            bool(abs(H[2, 0]) > 0.001 or abs(H[2, 1]) > 0.001))


def sequence(w=None):
    """A video that walks through every case: a rejected view, a frame next to it (the reference
    advances its state to the rejected frame, so this one is valid again), the way back (which
    breaks a rule again, or not: 1 / 1.4 is within the scale bounds) and a steady frame.  Returns
    (frames, label per frame after the first: the case name for the views that must be rejected,
    None elsewhere)."""
    w = world() if w is None else w
    frames, labels = [frame(w), frame(w, dx=2, dy=-2)], [None]
    for name, (_, kw) in CASES.items():
        near = dict(kw)
        near["dx"] = near.get("dx", 0) + 2
        for lab, k in ((name, kw), (None, near), (None, dict(dx=4, dy=-2)), (None, dict(dx=5, dy=-4))):
            frames.append(frame(w, **k))
            labels.append(lab)
    return frames, labels
