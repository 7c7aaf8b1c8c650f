"""Seeded synthetic clips (frames + first-frame trimap) in the reference's input format.

Format follows ``EvalDataset.get_data`` (reference dataset.py:857-920): frames are uint8 **BGR**
``[H, W, 3]``; the first-frame trimap is one-hot float ``[3, H, W]`` in channel order
``[bg, unknown, fg]`` (dataset.py:887-893); demo-style clips use ``a = 1`` (dataset.py:866-867).
Pure numpy so both boxes generate identical bytes.
"""
import numpy as np


def _upsample_bilinear(lo, H, W):
    """Separable bilinear (align_corners=True style) upsample of [h, w, c] -> [H, W, c], numpy only."""
    h, w, _ = lo.shape
    ys = np.linspace(0, h - 1, H)
    xs = np.linspace(0, w - 1, W)
    y0 = np.floor(ys).astype(np.int64).clip(0, h - 2)
    x0 = np.floor(xs).astype(np.int64).clip(0, w - 2)
    fy = (ys - y0)[:, None, None]
    fx = (xs - x0)[None, :, None]
    rows = lo[y0] * (1 - fy) + lo[y0 + 1] * fy
    return rows[:, x0] * (1 - fx) + rows[:, x0 + 1] * fx


def synthetic_clip(H, W, T, seed=0):
    """Returns (frames uint8 [T,H,W,3] BGR, trimap one-hot float32 [3,H,W] (bg,un,fg))."""
    rng = np.random.Generator(np.random.PCG64(seed))
    h, w = max(H // 8, 2), max(W // 8, 2)
    base = rng.uniform(0, 1, (h, w, 3))
    drift = rng.uniform(-1, 1, (h, w, 3))
    frames = np.empty((T, H, W, 3), np.uint8)
    yy, xx = np.mgrid[0:H, 0:W]
    for t in range(T):
        lo = np.clip(base + 0.02 * t * drift, 0, 1)
        img = _upsample_bilinear(lo, H, W)
        # a moving brighter disc so consecutive frames differ structurally, not only in tone
        cy, cx = H / 2 + 0.5 * t, W / 2 + 1.0 * t
        disc = ((yy - cy) ** 2 + (xx - cx) ** 2) < (H / 4) ** 2
        img = np.where(disc[..., None], 0.35 + 0.65 * img, 0.75 * img)
        frames[t] = np.floor(img * 255.0 + 0.5).astype(np.uint8)
    return frames, disc_trimap(H, W)


def disc_trimap(H, W):
    yy, xx = np.mgrid[0:H, 0:W]
    r = np.sqrt((yy - H / 2) ** 2 + (xx - W / 2) ** 2)
    fg = r < H / 4
    un = (r >= H / 4) & (r < H / 3)
    bg = ~(fg | un)
    return np.stack([bg, un, fg]).astype(np.float32)


def soft_alpha(H, W, t=0):
    """A soft ground-truth alpha (V108-style flow, ``tri_gt=None``): smooth disc edge in [0,1]."""
    yy, xx = np.mgrid[0:H, 0:W]
    r = np.sqrt((yy - H / 2 - 0.5 * t) ** 2 + (xx - W / 2 - 1.0 * t) ** 2)
    edge = H / 32.0 + 1.0
    a = np.clip((H / 3.5 - r) / edge + 0.5, 0, 1)
    # quantise like an 8-bit PNG alpha channel (dataset.py:863-864)
    return (np.floor(a * 255.0 + 0.5) / 255.0).astype(np.float32)


def train_batch(B, S, H, W, seed=0, radius=4):
    """A training-style batch (reference dataset conventions, models/alpha/model.py:189-196): B clips of S frames,
    a [B,S,1,H,W] soft GT alpha, fg / bg [B,S,3,H,W] float32 BGR 0..255, tri [B,S,3,H,W] one-hot GT trimap of EVERY frame
    ([bg, unknown, fg]; unknown = the soft band of the alpha dilated by ``radius`` pixels).  Pure numpy."""
    a = np.empty((B, S, 1, H, W), np.float32)
    fg = np.empty((B, S, 3, H, W), np.float32)
    bg = np.empty((B, S, 3, H, W), np.float32)
    tri = np.zeros((B, S, 3, H, W), np.float32)
    for b in range(B):
        f, _ = synthetic_clip(H, W, S, seed * 100 + b)
        g, _ = synthetic_clip(H, W, S, seed * 100 + 50 + b)
        for s in range(S):
            al = soft_alpha(H, W, s + 2 * b)
            a[b, s, 0] = al
            fg[b, s] = f[s].astype(np.float32).transpose(2, 0, 1)
            bg[b, s] = g[s][::-1, ::-1].astype(np.float32).transpose(2, 0, 1)
            soft = (al > 0) & (al < 1)
            un = np.zeros_like(soft)
            pad = np.pad(soft, radius)
            for dy in range(2 * radius + 1):
                for dx in range(2 * radius + 1):
                    un |= pad[dy:dy + H, dx:dx + W]
            tri[b, s, 1] = un
            tri[b, s, 2] = (al >= 1) & ~un
            tri[b, s, 0] = ~(un | ((al >= 1) & ~un))
    return a, fg, bg, tri
