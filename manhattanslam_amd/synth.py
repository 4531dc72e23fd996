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


# ------------------------------------------------------------------------------------------------
# Round 4: a furnished room (curved and small objects, depth edges, z^2 sensor noise, dropout blobs) and a
# dense-in-view live map ("~35 % inside the current frustum").
#
# A scene is the box room plus a list of solids; rays are cast in closed form (float64).  All ray functions take
# flat arrays: origins o [N, 3], directions d [N, 3] (camera z component 1, so t IS the z-depth) and return the
# smallest positive t with the outward surface normal at the hit.
# ------------------------------------------------------------------------------------------------
def clutter_scene(seed=5):
    """Deterministic furniture: spheres, vertical cylinders (axis = world y) and yawed boxes standing on the floor
    (y = +1.5 is down: camera_pose yaws about the downward y axis) or hanging in the room, all >= 1.2 m away from the
    camera circle (radius 0.5 m about the origin)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    spheres, cyls, boxes = [], [], []

    def ring(rmin, rmax):
        a = rng.uniform(0, 2 * np.pi); r = rng.uniform(rmin, rmax)
        return np.clip(r * np.cos(a), -2.6, 2.6), np.clip(r * np.sin(a), -2.1, 2.1)
    for _ in range(26):     # spheres, some resting on the floor, some floating at camera height
        x, z = ring(1.5, 2.7); r = rng.uniform(0.1, 0.4)
        y = ROOM[1] - r if rng.random() < 0.5 else rng.uniform(-0.5, 0.6)
        spheres.append((np.array([x, y, z]), r))
    for _ in range(16):     # pillars and lamp posts
        x, z = ring(1.6, 2.8); r = rng.uniform(0.05, 0.25)
        y1 = ROOM[1]; y0 = y1 - rng.uniform(0.6, 2.6)
        cyls.append((x, z, r, y0, y1))
    for _ in range(26):     # cupboards, tables, small boxes: (centre, half extents, yaw)
        x, z = ring(1.6, 2.9); hx, hy, hz = rng.uniform(0.08, 0.5), rng.uniform(0.1, 0.7), rng.uniform(0.08, 0.5)
        boxes.append((np.array([x, ROOM[1] - hy, z]), np.array([hx, hy, hz]), rng.uniform(0, np.pi)))
    return dict(spheres=spheres, cyls=cyls, boxes=boxes)


ROOM_ONLY = dict(spheres=[], cyls=[], boxes=[])


def cast(scene, o, d, subsets=None):
    """(t [N], outward normal [N, 3], object id [N]) of the first hit of rays o + t d; id 0..5 = room faces, then the solids.
    subsets: optional {solid number: indices of the only rays that can hit it} (a caller that knows the projection culls with it)."""
    o = np.asarray(o, np.float64); d = np.asarray(d, np.float64)
    n = len(d)
    with np.errstate(divide="ignore", invalid="ignore"):
        t1 = (ROOM[None, :] - o) / d
        t2 = (-ROOM[None, :] - o) / d
    tw = np.where(d > 0, t1, np.where(d < 0, t2, np.inf))
    ax = np.argmin(tw, axis=1)
    t = tw[np.arange(n), ax]
    nrm = np.zeros((n, 3))
    nrm[np.arange(n), ax] = -np.sign(d[np.arange(n), ax])          # pointing into the room
    oid = 2 * ax + (d[np.arange(n), ax] < 0)
    nid = 6

    O, D = o, d          # all rays; o, d below are the rays the current solid is tested against
    sel = None

    def choose():
        nonlocal o, d, sel, n
        sel = None if subsets is None else subsets.get(nid - 6)
        o, d = (O, D) if sel is None else (O[sel], D[sel])
        n = len(d)

    def take(tc, nc, valid):
        m = valid & (tc > 1e-6) & (tc < (t if sel is None else t[sel]))
        i = np.flatnonzero(m) if sel is None else sel[m]
        t[i] = tc[m]; nrm[i] = nc[m]; oid[i] = nid
    with np.errstate(divide="ignore", invalid="ignore"):
        for c, r in scene["spheres"]:
            choose()
            oc = o - c
            a = (d * d).sum(1); b = (oc * d).sum(1); cc = (oc * oc).sum(1) - r * r
            disc = b * b - a * cc
            tc = (-b - np.sqrt(np.maximum(disc, 0))) / a
            take(tc, (oc + d * tc[:, None]) / r, disc > 0)
            nid += 1
        for x, z, r, y0, y1 in scene["cyls"]:
            choose()
            ox, oz = o[:, 0] - x, o[:, 2] - z
            a = d[:, 0] ** 2 + d[:, 2] ** 2; b = ox * d[:, 0] + oz * d[:, 2]; cc = ox * ox + oz * oz - r * r
            disc = b * b - a * cc
            tc = (-b - np.sqrt(np.maximum(disc, 0))) / a
            yh = o[:, 1] + d[:, 1] * tc
            ns = np.stack([(ox + d[:, 0] * tc) / r, np.zeros(n), (oz + d[:, 2] * tc) / r], 1)
            take(tc, ns, (disc > 0) & (a > 0) & (yh >= y0) & (yh <= y1))
            tc = (y0 - o[:, 1]) / d[:, 1]                              # top cap (the bottom one stands on the floor)
            hx, hz = ox + d[:, 0] * tc, oz + d[:, 2] * tc
            take(tc, np.tile([0.0, -1.0, 0.0], (n, 1)), (hx * hx + hz * hz <= r * r) & (d[:, 1] > 0))
            nid += 1
        for c, hs, yaw in scene["boxes"]:
            choose()
            cy_, sy_ = np.cos(yaw), np.sin(yaw)
            R = np.array([[cy_, 0, sy_], [0, 1, 0], [-sy_, 0, cy_]])      # box -> world
            ob = (o - c) @ R; db = d @ R                                    # world -> box (R^T applied to row vectors)
            ta = (-hs[None, :] - ob) / db; tb = (hs[None, :] - ob) / db
            tn = np.minimum(ta, tb); tf = np.maximum(ta, tb)
            tn = np.where(np.isnan(tn), -np.inf, tn); tf = np.where(np.isnan(tf), np.inf, tf)
            axn = np.argmax(tn, axis=1)
            tnear = tn[np.arange(n), axn]; tfar = tf.min(axis=1)
            nb = np.zeros((n, 3)); nb[np.arange(n), axn] = -np.sign(db[np.arange(n), axn])
            take(tnear, nb @ R.T, tnear <= tfar)
            nid += 1
    return t, nrm, oid


def bounding_spheres(scene):
    """(centre, radius) of every solid, in cast()'s numbering."""
    out = [(c, r) for c, r in scene["spheres"]]
    out += [(np.array([x, 0.5 * (y0 + y1), z]), float(np.hypot(r, 0.5 * (y1 - y0)))) for x, z, r, y0, y1 in scene["cyls"]]
    out += [(c, float(np.linalg.norm(hs))) for c, hs, _ in scene["boxes"]]
    return out


def _pose_matrix(k):
    return camera_pose(k).reshape(4, 4).T.astype(np.float64)


def clutter_frame(k, w=640, h=480, intr=TUM1, seed=7, scene=None, noise_z2=0.0015, blobs=0.07, edge_dropout=0.5):
    """Keyframe k of the furnished room: (gray u8, depth f32 metres, membership i32 all -1, pose f32[16], object id map).
    Depth = exact z-depth + N(0, (noise_z2 z^2)^2) (structured-light noise grows with z^2: ~6 mm at 2 m) with `blobs` of the
    image knocked out in elliptic patches and `edge_dropout` of the pixels next to a depth discontinuity invalid, as real
    sensors do -- so superpixels straddle depth edges with partial data, curved patches and a few dozen valid pixels."""
    scene = clutter_scene() if scene is None else scene
    rng = np.random.Generator(np.random.PCG64(seed * 100003 + k + 77))
    pose = camera_pose(k)
    T = _pose_matrix(k)
    u, v = np.meshgrid(np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64))
    dc = np.stack([(u - intr["cx"]) / intr["fx"], (v - intr["cy"]) / intr["fy"], np.ones_like(u)], axis=2).reshape(-1, 3)
    dw = dc @ T[:3, :3].T
    # cull: a solid is only tested against the rays of the pixel rectangle its bounding sphere projects into
    Ti = np.linalg.inv(T)
    subsets = {}
    for j, (c, r) in enumerate(bounding_spheres(scene)):
        cc = Ti[:3, :3] @ c + Ti[:3, 3]
        if cc[2] + r < 0.05:
            subsets[j] = np.zeros(0, np.int64)
            continue
        if cc[2] - r < 0.05:
            continue                                       # straddles the camera plane: all rays
        zn = cc[2] - r
        x0 = (cc[0] - r) / (zn if cc[0] - r < 0 else cc[2] + r); x1 = (cc[0] + r) / (zn if cc[0] + r > 0 else cc[2] + r)
        y0 = (cc[1] - r) / (zn if cc[1] - r < 0 else cc[2] + r); y1 = (cc[1] + r) / (zn if cc[1] + r > 0 else cc[2] + r)
        us = sorted((x0 * intr["fx"] + intr["cx"], x1 * intr["fx"] + intr["cx"])); vs = sorted((y0 * intr["fy"] + intr["cy"], y1 * intr["fy"] + intr["cy"]))
        ua, ub = max(int(np.floor(us[0])) - 1, 0), min(int(np.ceil(us[1])) + 2, w)
        va, vb = max(int(np.floor(vs[0])) - 1, 0), min(int(np.ceil(vs[1])) + 2, h)
        subsets[j] = (np.arange(va, vb)[:, None] * w + np.arange(ua, ub)[None, :]).reshape(-1) if ua < ub and va < vb else np.zeros(0, np.int64)
    t, nrm, oid = cast(scene, np.broadcast_to(T[:3, 3], dw.shape), dw, subsets)
    hit = T[:3, 3][None, :] + dw * t[:, None]
    t = t.reshape(h, w); oid = oid.reshape(h, w)
    depth = t + rng.normal(0.0, 1.0, size=t.shape) * noise_z2 * t * t
    # dropout: elliptic blobs ...
    s = w / 640.0
    area = 0.0
    yy, xx = np.mgrid[0:h, 0:w]
    while area < blobs * w * h:
        bx, by = rng.uniform(0, w), rng.uniform(0, h); ra, rb = rng.uniform(3, 28) * s, rng.uniform(3, 28) * s
        m = ((xx - bx) / ra) ** 2 + ((yy - by) / rb) ** 2 <= 1.0
        depth[m] = 0.0
        area += np.pi * ra * rb
    # ... and a share of the pixels at depth discontinuities (> 5 cm to a 4-neighbour)
    edge = np.zeros_like(t, bool)
    dx = np.abs(np.diff(t, axis=1)) > 0.05; dy = np.abs(np.diff(t, axis=0)) > 0.05
    edge[:, :-1] |= dx; edge[:, 1:] |= dx; edge[:-1, :] |= dy; edge[1:, :] |= dy
    depth[edge & (rng.random(t.shape) < edge_dropout)] = 0.0
    # texture: a checker on the two dominant tangent coordinates of the hit, phase / contrast / cell size by object
    an = np.abs(nrm)
    dom = np.argmax(an, axis=1)
    a = np.where(dom == 0, hit[:, 1], hit[:, 0]); b = np.where(dom == 2, hit[:, 1], hit[:, 2])
    oidf = oid.reshape(-1)
    cell = np.where(oidf < 6, 0.25, 0.07 + 0.02 * (oidf % 5))
    chk = (np.floor(a / cell).astype(np.int64) + np.floor(b / cell).astype(np.int64)) & 1
    lo = np.where(oidf < 6, 60, 30 + (oidf * 37) % 90); hi = np.where(oidf < 6, 180, 140 + (oidf * 53) % 110)
    shade = 0.65 + 0.35 * np.abs((nrm * dw).sum(1)) / np.linalg.norm(dw, axis=1)          # Lambert-ish: curved objects get gradients
    gray = (np.where(chk == 1, hi, lo) * shade).reshape(h, w) + rng.integers(-4, 5, size=t.shape)
    member = np.full(((h + 1) // 2, (w + 1) // 2), -1, np.int32)
    return (np.clip(np.rint(gray), 0, 255).astype(np.uint8), np.ascontiguousarray(depth.astype(np.float32)), member, pose, oid.astype(np.int32))


def surfel_map_dense(n, ref=0, seed=11, w=640, h=480, intr=TUM1, scene=None, k_lo=-150, k_hi=214, flip=0.01, floating=0.003,
                     min_update_times=5, order="creation", scatter=0.01):
    """n live surfels the way a camera that panned over keyframes k_lo .. k_hi - 1 would have left them: each surfel is the
    back-projection of a uniformly drawn pixel of a uniformly drawn source keyframe onto the scene.  With the 0.5 degree pan of
    camera_pose() and TUM1's 63.5 degree horizontal field of view a frustum overlaps the sources within +-127 keyframes, so for the
    default 364 sources about 127 / 364 = 35 % of the map lies inside the frustum of ANY keyframe 0..63 (SURVEY.md 8(d) config 3;
    the reference's local map = the surfels of the last <= 10 pose-graph hops, src/SurfelMapping.cpp:326-351, is mostly in view).
    order="creation": array order = (source keyframe, superpixel raster index), the order initializeSurfels appends them in
    (src/SurfelFusion.cpp:285-331); "random": no locality between neighbours in the array at all.
    flip / floating: shares of surfels with an outward normal (normal-disagreement deletion, :230-233) and of surfels floating
    1.2-2 m in front of the surface (occlusion deletion, :208-211)."""
    scene = ROOM_ONLY if scene is None else scene
    rng = np.random.Generator(np.random.PCG64(seed))
    ks = rng.integers(k_lo, k_hi, n)
    u = rng.uniform(1.0, w - 2.0, n); v = rng.uniform(1.0, h - 2.0, n)
    if order == "creation":
        key = (ks - k_lo).astype(np.int64) * ((w // 8) * (h // 8)) + (v.astype(np.int64) // 8) * (w // 8) + u.astype(np.int64) // 8
        o_ = np.argsort(key, kind="stable")
        ks, u, v = ks[o_], u[o_], v[o_]
    Ts = {k: _pose_matrix(k) for k in range(k_lo, k_hi)}
    Rm = np.stack([Ts[k][:3, :3] for k in range(k_lo, k_hi)]); Cm = np.stack([Ts[k][:3, 3] for k in range(k_lo, k_hi)])
    dc = np.stack([(u - intr["cx"]) / intr["fx"], (v - intr["cy"]) / intr["fy"], np.ones(n)], axis=1)
    R = Rm[ks - k_lo]; cw = Cm[ks - k_lo]
    dw = np.einsum("nij,nj->ni", R, dc)
    t, nrm, _ = cast(scene, cw, dw)
    far = rng.random(n) < floating
    shift = np.where(far, np.minimum(rng.uniform(1.2, 2.0, n), 0.7 * t), 0.0)
    p = cw + dw * (t - shift)[:, None] + nrm * rng.normal(0, scatter, n)[:, None]
    fl = rng.random(n) < flip
    nv = nrm * np.where(fl, -1.0, 1.0)[:, None] + rng.normal(0, 0.05, size=(n, 3))
    nv /= np.linalg.norm(nv, axis=1, keepdims=True)
    m = np.zeros(n, surfel_dtype())
    m["px"], m["py"], m["pz"] = p[:, 0], p[:, 1], p[:, 2]
    m["nx"], m["ny"], m["nz"] = nv[:, 0], nv[:, 1], nv[:, 2]
    m["size"] = rng.uniform(0.005, 0.03, n)
    m["color"] = rng.integers(0, 256, n)
    m["r"] = m["g"] = m["b"] = rng.integers(0, 256, n)
    m["weight"] = rng.uniform(1, 20, n)
    m["updateTimes"] = rng.integers(min_update_times, 21, n)
    m["lastUpdate"] = ref - rng.integers(0, 9, n)
    return m


MOVING_STEP = 8        # "moving" regime: keyframe f looks along camera_pose(8 f): a 4 degree pan per keyframe, 256 degrees over 64 keyframes
MOVING_PERIOD = 10.0   # degrees of azimuth: the pre-seeded map lacks MOVING_HOLE degrees of every MOVING_PERIOD (ahead of the first view)
MOVING_HOLE = 4.0


def surfel_map_moving(n, ref=0, seed=11, w=640, h=480, intr=TUM1, scene=None, n_keyframes=64, **kw):
    """The pre-seeded map of the moving-camera regime: surfel_map_dense over the whole sweep of keyframes MOVING_STEP * (0 .. n_keyframes - 1)
    (so that about a third of it is in view all along), with vertical stripes of the room left UNMAPPED ahead of the first view: the surfels
    whose azimuth about the room centre falls into the first MOVING_HOLE degrees of every MOVING_PERIOD are removed.  A pan then meets what a
    real one meets (src/SurfelFusion.cpp:285-331, src/SurfelMapping.cpp:366-391): every keyframe the strip entering the image holds
    superpixels that no surfel projects into -- they spawn new surfels, a hundred or more per keyframe --, and the flipped / floating
    surfels of the mapped stripes that enter are deleted, about as many; holes are refilled, the tail moves, the array grows."""
    k_hi = MOVING_STEP * (n_keyframes - 1) + 128
    m = surfel_map_dense(int(n * MOVING_PERIOD / (MOVING_PERIOD - 0.8 * MOVING_HOLE)), ref=ref, seed=seed, w=w, h=h, intr=intr, scene=scene, k_lo=-150, k_hi=k_hi, **kw)
    az = np.rad2deg(np.arctan2(m["px"].astype(np.float64), m["pz"].astype(np.float64)))     # camera_pose(k) looks along azimuth 0.5 k degrees
    first_view = 0.5 * np.rad2deg(2 * np.arctan(0.5 * w / abs(intr["fx"])))
    ahead = (az - first_view) % 360.0
    hole = ((ahead % MOVING_PERIOD) < MOVING_HOLE) & (ahead < 0.5 * k_hi + 40.0) & (((az + 180.0) % 360.0 - 180.0) > first_view)
    m = m[~hole]
    return m[:n] if len(m) >= n else m


def in_view_fraction(m, k, w=640, h=480, intr=TUM1, near=0.5, far=30.0):
    """Share of the surfels m that keyframe k's fuse step finds in range and inside the image (src/SurfelFusion.cpp:196-207)."""
    T = np.linalg.inv(_pose_matrix(k))
    p = np.stack([m["px"], m["py"], m["pz"]], 1).astype(np.float64) @ T[:3, :3].T + T[:3, 3]
    with np.errstate(divide="ignore", invalid="ignore"):
        pu = np.floor(p[:, 0] * intr["fx"] / p[:, 2] + intr["cx"] + 0.5); pv = np.floor(p[:, 1] * intr["fy"] / p[:, 2] + intr["cy"] + 0.5)
    ok = (p[:, 2] >= near) & (p[:, 2] <= far) & (pu >= 1) & (pu <= w - 2) & (pv >= 1) & (pv <= h - 2)
    return float(ok.mean())


def sequence_seeds(rank):
    """Seeds of rank r's independent sequence ('seeds offset by rank')."""
    return {"frame": 7 + 1000 * rank, "orb": ORB_SEED + 1000 * rank, "map": 11 + rank}


def bench_inputs(rank, D, n_surfels, W, H, intr, variant="A", dropout=0.02, need_orb_texture=True, map_kind="dense", map_order="creation",
                 scene="room", flip=0.01, floating=0.003):
    """The benchmark's workload, shared by bench.py and tools/cpu_baseline.py so that the GPU and the CPU baseline see the same bytes:
    D distinct RGB-D frames of this rank's sequence + the pre-seeded live map (numpy, host).
    map_kind "dense": surfel_map_dense (~35 % of the map inside the frustum of every keyframe: SURVEY.md 8(d) config 3 as written);
    "sparse": the area-uniform room map of rounds 1-3 (~6 % in view).  scene "room": the bare box room; "clutter": the furnished room."""
    sd = sequence_seeds(rank)
    sc = clutter_scene() if scene == "clutter" else ROOM_ONLY
    grays, depths, poses = [], [], []
    member = None
    kstep = MOVING_STEP if map_kind == "moving" else 1
    for f in range(D):
        if scene == "clutter":
            g, depth, member, pose, _ = clutter_frame(kstep * f, w=W, h=H, intr=intr, seed=sd["frame"], scene=sc)
        else:
            g, depth, member, pose = surfel_frame(kstep * f, w=W, h=H, intr=intr, variant=variant, seed=sd["frame"], dropout=dropout)
        # one gray image per frame, used by both stages: the textured ORB frame (the wall checker alone has too few corners)
        grays.append(orb_frame(sd["orb"] + f, W, H) if need_orb_texture else g)
        depths.append(depth)
        poses.append(pose)
    if map_kind == "moving":
        # (three times the flipped / floating shares of the stationary map: the deletions of a keyframe come from the mapped stripes that ENTER the view)
        smap = surfel_map_moving(n_surfels, ref=0, seed=sd["map"], w=W, h=H, intr=intr, scene=sc, n_keyframes=D, flip=3 * flip, floating=3 * floating, min_update_times=5)
    elif map_kind == "dense":
        smap = surfel_map_dense(n_surfels, ref=0, seed=sd["map"], w=W, h=H, intr=intr, scene=sc, flip=flip, floating=floating, min_update_times=5,
                                order=map_order)
    else:
        smap = surfel_map(n_surfels, ref=0, seed=sd["map"], min_update_times=5)
    return np.stack(grays), np.stack(depths), member, poses, smap
