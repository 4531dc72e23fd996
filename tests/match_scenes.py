"""Synthetic inputs for the Hamming-matching tests (shared by the CPU oracle tests and the GPU parity tests)."""
import numpy as np

KEYPOINT_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                           ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
SCALE = np.cumprod(np.concatenate([[1.0], np.full(7, 1.2)]).astype(np.float32)).astype(np.float32)


def params(oracle_lib_or_none, th, check_orientation=True, w=640, h=480, fx=517.3, fy=516.5, cx=318.6, cy=255.3, bf=40.0, dtype=None):
    p = np.zeros(1, dtype)
    p["fx"], p["fy"], p["cx"], p["cy"], p["bf"] = fx, fy, cx, cy, bf
    p["minX"], p["maxX"], p["minY"], p["maxY"] = 0.0, float(w), 0.0, float(h)     # ComputeImageBounds without distortion
    p["th"] = th
    p["check_orientation"] = 1 if check_orientation else 0
    p["nlevels"] = 8
    p["scale_factors"][0, :8] = SCALE
    return p


def grid_cells(un_xy, p):
    """Frame::PosInGrid (src/Frame.cc:418-427) in float32: posX * 48 + posY or -1."""
    minX, maxX, minY, maxY = (np.float32(p[k][0]) for k in ("minX", "maxX", "minY", "maxY"))
    wInv = np.float32(64) / (maxX - minX); hInv = np.float32(48) / (maxY - minY)
    px = np.round((un_xy[:, 0] - minX) * wInv).astype(np.int64); py = np.round((un_xy[:, 1] - minY) * hInv).astype(np.int64)
    ok = (px >= 0) & (px < 64) & (py >= 0) & (py < 48)
    return np.where(ok, px * 48 + py, -1).astype(np.int32)


def random_pair(seed, p, n_cur=900, n_last=850, shift=(6.0, -4.0), tz=0.0, z=2.0, desc_noise=12, obs_frac=0.7, cluster=False):
    """A current frame of n_cur random keypoints and n_last last-frame map points, most of which reproject next to a current
    keypoint with a similar descriptor (so windows hold true matches plus random competitors)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    fx, fy, cx, cy = (float(p[k][0]) for k in ("fx", "fy", "cx", "cy"))
    W, H = float(p["maxX"][0]), float(p["maxY"][0])
    kps = np.zeros(n_cur, KEYPOINT_DTYPE)
    if cluster:   # everything inside a small region: dozens of candidates per window
        xy = np.stack([rng.uniform(200, 330, n_cur), rng.uniform(150, 250, n_cur)], 1)
    else:
        xy = np.stack([rng.uniform(16, W - 16, n_cur), rng.uniform(16, H - 16, n_cur)], 1)
    xy = np.round(xy).astype(np.float32)   # level-0-like integer coordinates (several keypoints may share a pixel: ties)
    kps["x"], kps["y"] = xy[:, 0], xy[:, 1]
    kps["octave"] = rng.integers(0, 8, n_cur)
    kps["angle"] = rng.uniform(0, 360, n_cur).astype(np.float32)
    kps["class_id"] = -1
    desc = rng.integers(0, 256, (n_cur, 32), dtype=np.uint8)
    depth = np.where(rng.random(n_cur) < 0.85, z + rng.uniform(-0.3, 0.3, n_cur), -1.0).astype(np.float32)
    uright = np.where(depth > 0, xy[:, 0] - np.float32(p["bf"][0]) / np.where(depth > 0, depth, 1), -1.0).astype(np.float32)
    cur = dict(kps=kps, un_xy=xy.copy(), uright=uright, grid_cell=grid_cells(xy, p), desc=desc)
    # last-frame points: perturbed copies of current keypoints (true matches) + unrelated ones
    src = rng.integers(0, n_cur, n_last)
    unrelated = rng.random(n_last) < 0.15
    # target projection in the CURRENT frame: next to keypoint src; world point = current-camera point - t (R = I)
    u = xy[src, 0] + rng.normal(0, 1.5, n_last); v = xy[src, 1] + rng.normal(0, 1.5, n_last)
    zz = np.where(depth[src] > 0, depth[src], z).astype(np.float64)
    t = np.array([shift[0] * z / fx, shift[1] * z / fy, tz])
    xyz = (np.stack([(u - cx) * zz / fx, (v - cy) * zz / fy, zz], 1) - t[None, :]).astype(np.float32)
    ld = desc[src].copy()
    flip = rng.integers(0, 256, (n_last, desc_noise))
    for k in range(desc_noise):
        ld[np.arange(n_last), flip[:, k] // 8] ^= (1 << (flip[:, k] % 8)).astype(np.uint8)
    ld[unrelated] = rng.integers(0, 256, (int(unrelated.sum()), 32), dtype=np.uint8)
    octave = np.clip(kps["octave"][src] + rng.integers(-1, 2, n_last), 0, 7).astype(np.int32)
    angle = (kps["angle"][src] + np.where(rng.random(n_last) < 0.8, rng.normal(3.0, 2.0, n_last), rng.uniform(0, 360, n_last))) % 360
    flags = ((rng.random(n_last) < 0.92).astype(np.uint8)) | ((rng.random(n_last) < obs_frac).astype(np.uint8) << 1)
    xyz[rng.random(n_last) < 0.01, 2] *= -1          # a few points behind the camera (invzc < 0)
    xyz[rng.random(n_last) < 0.005, 2] = 0.0         # and a few at depth 0 (inf / NaN projection)
    last = dict(xyz=xyz, desc=ld, flags=flags, octave=octave, angle=angle.astype(np.float32))
    Tl = np.eye(4, dtype=np.float32)
    Tc = np.eye(4, dtype=np.float32)
    Tc[0, 3] = shift[0] * z / fx; Tc[1, 3] = shift[1] * z / fy; Tc[2, 3] = tz   # x_c = x_w + t: pixels move by about +shift
    return cur, last, Tc, Tl
