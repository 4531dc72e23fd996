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
    n_small = int(9500 * s * s)
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
    # ... whose top-left 64 x 64 corner holds contrasts of at most 9 + 9 < 20 only: at least one FAST cell lies inside it and is
    # guaranteed to need the 20 -> 7 fallback (the +-12 rectangles above usually produce such cells, but not for every seed)
    ss = int(64 * s)
    img[ly:ly + ss, lx:lx + ss] = 110
    for iy, gy in enumerate(range(ly + 4, ly + ss - 12, 16)):      # isolated 8 x 8 squares, +-9 against the base
        for ix, gx in enumerate(range(lx + 4, lx + ss - 12, 16)):
            sg = 1 if (ix + iy) & 1 else -1
            img[gy:gy + 8, gx:gx + 8] = 110 + 9 * sg
            for cy, cx in ((gy, gx), (gy, gx + 7), (gy + 7, gx), (gy + 7, gx + 7)):   # a stronger corner pixel: on a perfectly uniform square
                img[cy, cx] = 110 + 10 * sg                                        # all scores tie and the strict 3x3 NMS removes every corner
    # flat patch
    fx0, fy0, fs = int(420 * s), int(60 * s), int(96 * s)
    img[fy0:fy0 + fs, fx0:fx0 + fs] = 128
    return img


def orb_frames(n, seed=ORB_SEED, w=640, h=480):
    return np.stack([orb_frame(seed + i, w, h) for i in range(n)])


# ------------------------------------------------------------------------------------------------
# Surfel-fusion scene: box room, camera on a circular pan.
# ------------------------------------------------------------------------------------------------
TUM1 = dict(fx=517.306408, fy=516.469215, cx=318.643040, cy=255.313989)   # Example/TUM1.yaml:8-11
ICL = dict(fx=481.20, fy=-480.00, cx=319.50, cy=239.50)                   # Example/ICL.yaml:8-11


def scaled_intrinsics(intr, w):
    """Intrinsics of the same camera at image width w (the 1280x960 sequences of BASELINE config 5 use 2x TUM1)."""
    s = w / 640.0
    return {k: v * s for k, v in intr.items()}


ROOM = np.array([3.0, 1.5, 2.5])  # half extents: x in [-3,3], y in [-1.5,1.5], z in [-2.5,2.5]

SURFEL_FIELDS = ("px", "py", "pz", "nx", "ny", "nz", "size", "color", "r", "g", "b", "weight", "updateTimes", "lastUpdate")


def surfel_dtype():
    return np.dtype([(n, "<i4" if n in ("r", "g", "b", "updateTimes", "lastUpdate") else "<f4") for n in SURFEL_FIELDS])


def camera_pose(k):
    """Twc of keyframe k as a column-major float32[16] (Eigen::Matrix4f storage)."""
    th = np.deg2rad(0.5 * k)
    c, s = np.cos(th), np.sin(th)
    T = np.eye(4)
    T[:3, :3] = np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])       # yaw about the (downward) y axis
    T[:3, 3] = [0.5 * np.cos(th), 0.1, 0.5 * np.sin(th)]
    return np.ascontiguousarray(T.T.astype(np.float32).reshape(16))  # column-major


def _ray_box(cw, dw):
    """Distance t >= 0 along rays dw from the interior point cw to the room walls; also the hit axis."""
    with np.errstate(divide="ignore", invalid="ignore"):
        t1 = (ROOM[None, None, :] - cw) / dw
        t2 = (-ROOM[None, None, :] - cw) / dw
    t = np.where(dw > 0, t1, np.where(dw < 0, t2, np.inf))
    axis = np.argmin(t, axis=2)
    return np.min(t, axis=2), axis


def surfel_frame(k, w=640, h=480, intr=TUM1, variant="A", seed=7, dropout=0.02):
    """(gray u8 [h,w], depth f32 [h,w] metres, membership i32 [h/2,w/2], pose f32[16]) for keyframe k."""
    rng = np.random.Generator(np.random.PCG64(seed * 100003 + k))
    pose = camera_pose(k)
    T = pose.reshape(4, 4).T.astype(np.float64)
    u, v = np.meshgrid(np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64))
    dc = np.stack([(u - intr["cx"]) / intr["fx"], (v - intr["cy"]) / intr["fy"], np.ones_like(u)], axis=2)
    dw = dc @ T[:3, :3].T
    cw = T[:3, 3][None, None, :]
    t, axis = _ray_box(cw, dw)
    hit = cw + dw * t[:, :, None]
    depth = t + rng.uniform(-0.002, 0.002, size=t.shape)
    depth[rng.random(t.shape) < dropout] = 0.0   # invalid pixels (dropout = 0.02 unless a test wants cleaner depth)
    # wall checker: the two tangent coordinates of the hit wall
    a = np.where(axis == 0, hit[:, :, 1], hit[:, :, 0])
    b = np.where(axis == 2, hit[:, :, 1], hit[:, :, 2])
    chk = (np.floor(a / 0.25).astype(np.int64) + np.floor(b / 0.25).astype(np.int64)) & 1
    gray = np.where(chk == 1, 180, 60) + rng.integers(-4, 5, size=t.shape)
    member = np.full(((h + 1) // 2, (w + 1) // 2), -1, np.int32)
    if variant == "B":
        s = w / 640.0
        for pid, (x0, y0, x1, y1) in enumerate(((20, 20, 90, 70), (130, 100, 210, 160), (240, 30, 300, 220))):
            member[int(y0 * s):int(y1 * s), int(x0 * s):int(x1 * s)] = pid
    return (np.clip(gray, 0, 255).astype(np.uint8), np.ascontiguousarray(depth.astype(np.float32)), member, pose)


def surfel_map(n, ref=0, seed=11, min_update_times=1):
    """n live surfels pre-seeded on the room surfaces (area-uniform), as a structured array."""
    rng = np.random.Generator(np.random.PCG64(seed))
    ex = ROOM * 2
    areas = np.array([ex[1] * ex[2], ex[1] * ex[2], ex[0] * ex[2], ex[0] * ex[2], ex[0] * ex[1], ex[0] * ex[1]])
    face = rng.choice(6, size=n, p=areas / areas.sum())
    p = rng.uniform(-1, 1, size=(n, 3)) * ROOM[None, :]
    ax = face // 2
    sign = np.where(face % 2 == 0, 1.0, -1.0)
    off = rng.normal(0, 0.01, size=n)                      # 1 cm scatter about the wall
    far = rng.random(n) < 0.03                             # 3 %: floating well inside the room (occlusion test)
    off = np.where(far, -rng.uniform(1.2, 2.0, size=n), off)
    p[np.arange(n), ax] = sign * (ROOM[ax] + off)
    nrm = np.zeros((n, 3))
    flip = rng.random(n) < 0.1                             # 10 % point outward -> normal-disagreement deletion
    nrm[np.arange(n), ax] = -sign * np.where(flip, -1.0, 1.0)
    nrm += rng.normal(0, 0.05, size=(n, 3))
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    m = np.zeros(n, surfel_dtype())
    m["px"], m["py"], m["pz"] = p[:, 0], p[:, 1], p[:, 2]
    m["nx"], m["ny"], m["nz"] = nrm[:, 0], nrm[:, 1], nrm[:, 2]
    m["size"] = rng.uniform(0.005, 0.03, n)
    m["color"] = rng.integers(0, 256, n)
    m["r"] = m["g"] = m["b"] = rng.integers(0, 256, n)
    m["weight"] = rng.uniform(1, 20, n)
    m["updateTimes"] = rng.integers(min_update_times, 21, n)
    m["lastUpdate"] = ref - rng.integers(0, 9, n)
    return m


def depth_u16(depth_m, factor=5000.0):
    """Metres -> the raw 16-bit depth image the plane extractor reads (TUM / ICL convention: 5000 units per metre)."""
    return np.clip(np.rint(depth_m * factor), 0, 65535).astype(np.uint16)
