"""Seeded synthetic panned video with ground-truth homographies.

World texture: five octaves of value noise (40..215) per channel plus random
filled rectangles and discs (FAST corners), clamped to [1, 255] so that
``pixel > 0`` equals the frame footprint.  Camera path: frame t shows the world
through G_t = T(cx + 2t + j, cy - 2t + j) . R(0.02 deg * t) . S(1 + 1e-4 t),
i.e. frame_t(p) = world(G_t p); about 3 px of motion per frame, no perspective.
The homography that maps frame t onto frame t-1 is G_{t-1}^-1 G_t.
"""
import numpy as np


def make_world(h, w, seed=0, n_shapes=None):
    rng = np.random.default_rng(seed)
    img = np.zeros((h, w, 3), np.float32)
    for c in range(3):
        acc = np.zeros((h, w), np.float32)
        amp, tot = 1.0, 0.0
        for o in range(5):
            cell = 256 >> o
            gh, gw = h // cell + 2, w // cell + 2
            g = rng.random((gh, gw), dtype=np.float32)
            ys = np.arange(h, dtype=np.float32) / cell
            xs = np.arange(w, dtype=np.float32) / cell
            y0, x0 = ys.astype(np.int32), xs.astype(np.int32)
            fy, fx = (ys - y0)[:, None], (xs - x0)[None, :]
            a = g[y0][:, x0]
            b = g[y0][:, x0 + 1]
            cc = g[y0 + 1][:, x0]
            d = g[y0 + 1][:, x0 + 1]
            acc += amp * ((a * (1 - fx) + b * fx) * (1 - fy) + (cc * (1 - fx) + d * fx) * fy)
            tot += amp
            amp *= 0.5
        img[:, :, c] = 40 + 175 * acc / tot
    srng = np.random.default_rng(seed + 1)
    if n_shapes is None:
        n_shapes = max(16, (h * w) // 840)
    for _ in range(n_shapes):
        x, y = int(srng.integers(0, w)), int(srng.integers(0, h))
        col = srng.integers(1, 256, 3).astype(np.float32)
        if srng.random() < 0.5:
            rw, rh = int(srng.integers(4, 40)), int(srng.integers(4, 40))
            img[y:y + rh, x:x + rw] = col
        else:
            r = int(srng.integers(3, 20))
            ya, yb = max(0, y - r), min(h, y + r + 1)
            xa, xb = max(0, x - r), min(w, x + r + 1)
            yy, xx = np.mgrid[ya:yb, xa:xb]
            m = (yy - y) ** 2 + (xx - x) ** 2 <= r * r
            img[ya:yb, xa:xb][m] = col
    return np.clip(np.rint(img), 1, 255).astype(np.uint8)


def camera_matrix(t, cx, cy, jitter=(0.0, 0.0), step=2.0, rot_deg=0.02, zoom=1e-4):
    th = np.deg2rad(rot_deg * t)
    s = 1.0 + zoom * t
    c, sn = np.cos(th) * s, np.sin(th) * s
    return np.array([[c, -sn, cx + step * t + jitter[0]],
                     [sn, c, cy - step * t + jitter[1]],
                     [0.0, 0.0, 1.0]])


def camera_path(n_frames, cx, cy, seed=2, **kw):
    rng = np.random.default_rng(seed)
    out = []
    for t in range(n_frames):
        j = rng.uniform(-0.5, 0.5, 2) if t > 0 else (0.0, 0.0)
        out.append(camera_matrix(t, cx, cy, j, **kw))
    return out


def bounded_path(n_frames, fh, fw, seed=2, margin=150):
    """Camera path for long videos that keeps every frame inside a canvas of 2 fh x 1.2 fw whose
    first frame sits at the bottom centre (the reference's layout, main.py:80-90): the camera
    climbs and descends over 0.8 fh (triangular wave, 2.5 px/frame), sways +-0.07 fw sideways,
    rotates +-0.5 degrees and zooms +-1 %, plus +-0.5 px of jitter.  Returns (G list, world
    height, world width) with frame_t(p) = world(G_t p)."""
    rng = np.random.default_rng(seed)
    ay, ax = 0.8 * fh, 0.07 * fw
    wh, ww = int(fh + ay + 2 * margin), int(fw + 2 * ax + 2 * margin)
    cx, cy = margin + ax, wh - fh - margin
    out = []
    for t in range(n_frames):
        j = rng.uniform(-0.5, 0.5, 2) if t > 0 else (0.0, 0.0)
        u = 2.5 * t / ay
        tri = 2.0 * abs(u / 2.0 - np.floor(u / 2.0 + 0.5))           # 0 -> 1 -> 0, period 2
        th = np.deg2rad(0.5 * np.sin(2 * np.pi * t / 400.0))
        s = 1.0 + 0.01 * np.sin(2 * np.pi * t / 700.0)
        c, sn = np.cos(th) * s, np.sin(th) * s
        out.append(np.array([[c, -sn, cx + ax * np.sin(2 * np.pi * t / 360.0) + j[0]],
                             [sn, c, cy - ay * tri + j[1]],
                             [0.0, 0.0, 1.0]]))
    return out, wh, ww


def render_frame(world, G, fh, fw):
    """frame(p) = bilinear world(G p), u8, clamped to [1, 255] inside the world."""
    H, W = world.shape[:2]
    ys, xs = np.mgrid[0:fh, 0:fw].astype(np.float64)
    d = G[2, 0] * xs + G[2, 1] * ys + G[2, 2]
    u = (G[0, 0] * xs + G[0, 1] * ys + G[0, 2]) / d
    v = (G[1, 0] * xs + G[1, 1] * ys + G[1, 2]) / d
    u0 = np.clip(np.floor(u).astype(np.int64), 0, W - 2)
    v0 = np.clip(np.floor(v).astype(np.int64), 0, H - 2)
    a = np.clip(u - u0, 0, 1)[..., None].astype(np.float32)
    b = np.clip(v - v0, 0, 1)[..., None].astype(np.float32)
    w = world.astype(np.float32)
    p = (w[v0, u0] * (1 - a) + w[v0, u0 + 1] * a) * (1 - b) + (w[v0 + 1, u0] * (1 - a) + w[v0 + 1, u0 + 1] * a) * b
    return np.clip(np.rint(p), 1, 255).astype(np.uint8)


def make_video(fh, fw, n_frames, world=None, world_size=None, seed=0):
    """Returns (frames list, G list).  The world defaults to a square big enough for the path."""
    if world is None:
        if world_size is None:
            side = int(max(fh, fw) + 4 * n_frames + 256)
            world_size = (side, side)
        world = make_world(world_size[0], world_size[1], seed)
    H, W = world.shape[:2]
    cx = 64.0
    cy = H - fh - 64.0
    Gs = camera_path(n_frames, cx, cy, seed + 2)
    return [render_frame(world, G, fh, fw) for G in Gs], Gs


def relative_h(G_prev, G_cur):
    """Homography taking frame-cur pixel coordinates to frame-prev pixel coordinates."""
    Hr = np.linalg.inv(G_prev) @ G_cur
    return Hr / Hr[2, 2]


def render_video_torch(world, Gs, fh, fw, device):
    """All frames of a camera path rendered on the GPU (float32 bilinear, rounded, clamped to
    [1, 255]); returns a uint8 tensor [T, fh, fw, 3] resident on `device`."""
    import torch
    wt = torch.from_numpy(world).to(device).float()
    H, W = world.shape[:2]
    ys, xs = torch.meshgrid(torch.arange(fh, device=device, dtype=torch.float32),
                            torch.arange(fw, device=device, dtype=torch.float32), indexing="ij")
    out = torch.empty((len(Gs), fh, fw, 3), dtype=torch.uint8, device=device)
    for t, G in enumerate(Gs):
        g = [[float(v) for v in row] for row in G]
        d = g[2][0] * xs + g[2][1] * ys + g[2][2]
        u = (g[0][0] * xs + g[0][1] * ys + g[0][2]) / d
        v = (g[1][0] * xs + g[1][1] * ys + g[1][2]) / d
        u0 = u.floor().clamp(0, W - 2)
        v0 = v.floor().clamp(0, H - 2)
        a = (u - u0).clamp(0, 1)[..., None]
        b = (v - v0).clamp(0, 1)[..., None]
        iu, iv = u0.long(), v0.long()
        p = (wt[iv, iu] * (1 - a) + wt[iv, iu + 1] * a) * (1 - b) + (wt[iv + 1, iu] * (1 - a) + wt[iv + 1, iu + 1] * a) * b
        out[t] = p.round().clamp(1, 255).to(torch.uint8)
    return out
