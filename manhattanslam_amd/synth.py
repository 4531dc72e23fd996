"""Deterministic synthetic inputs shared by the tests, smoke() and bench.py.

Pure numpy (PCG64 streams), so the GPU path and the CPU oracle see byte-identical frames.
"""
import numpy as np

ORB_SEED = 20210530


def _value_noise(rng, h, w, cell, amp):
    gh, gw = h // cell + 2, w // cell + 2
    g = rng.uniform(-amp, amp, size=(gh, gw)).astype(np.float32)
    ys = (np.arange(h, dtype=np.float32) / cell)
    xs = (np.arange(w, dtype=np.float32) / cell)
    y0 = ys.astype(np.int32); x0 = xs.astype(np.int32)
    fy = (ys - y0)[:, None]; fx = (xs - x0)[None, :]
    a = g[y0][:, x0]; b = g[y0][:, x0 + 1]; c = g[y0 + 1][:, x0]; d = g[y0 + 1][:, x0 + 1]
    return (a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy


def orb_frame(seed=ORB_SEED, w=640, h=480):
    """Textured 8-bit frame: 3-octave value noise + 600 rectangles + one flat patch (empty FAST cells)
    + one low-contrast patch (cells that need the 20 -> 7 threshold fallback)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    img = np.full((h, w), 128.0, np.float32)
    for cell, amp in ((32, 40.0), (16, 20.0), (8, 10.0)):
        img += _value_noise(rng, h, w, cell, amp)
    img = np.clip(np.rint(img), 0, 255).astype(np.uint8)
    s = max(w, h) / 640.0
    n_rect = int(600 * s * s)
    xs = rng.integers(0, w, n_rect); ys = rng.integers(0, h, n_rect)
    ws = rng.integers(int(6 * s), int(60 * s) + 1, n_rect); hs = rng.integers(int(6 * s), int(60 * s) + 1, n_rect)
    gs = rng.integers(0, 256, n_rect)
    for x, y, rw, rh, g in zip(xs, ys, ws, hs, gs):
        img[y:y + rh, x:x + rw] = g
    # fine structure: many small high-contrast blocks so that level 0 yields thousands of FAST corners
    n_small = int(7000 * s * s)
    xs = rng.integers(0, w, n_small); ys = rng.integers(0, h, n_small)
    ws = rng.integers(2, 10, n_small); hs = rng.integers(2, 10, n_small)
    gs = rng.integers(0, 256, n_small)
    for x, y, rw, rh, g in zip(xs, ys, ws, hs, gs):
        img[y:y + rh, x:x + rw] = g
    # low-contrast patch: base 110, rectangles within +-12
    lx, ly, lw, lh = int(40 * s), int(300 * s), int(160 * s), int(120 * s)
    img[ly:ly + lh, lx:lx + lw] = 110
    for _ in range(int(60 * s * s)):
        x = lx + int(rng.integers(0, lw - 8)); y = ly + int(rng.integers(0, lh - 8))
        rw = int(rng.integers(4, int(30 * s))); rh = int(rng.integers(4, int(30 * s)))
        img[y:min(y + rh, ly + lh), x:min(x + rw, lx + lw)] = 110 + int(rng.integers(-12, 13))
    # flat patch
    fx0, fy0, fs = int(420 * s), int(60 * s), int(96 * s)
    img[fy0:fy0 + fs, fx0:fx0 + fs] = 128
    return img


def orb_frames(n, seed=ORB_SEED, w=640, h=480):
    return np.stack([orb_frame(seed + i, w, h) for i in range(n)])
