"""ctypes access to oracle/libmsl_oracle.so -- the CPU checker.  Test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ODIR = os.path.join(ROOT, "oracle")
OLIB = os.environ.get("MSL_ORACLE_LIB", os.path.join(ODIR, "libmsl_oracle.so"))   # MSL_ORACLE_LIB: the blur-variant oracle (tests only)

KEYPOINT_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                           ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
SURFEL_DTYPE = np.dtype([("px", "<f4"), ("py", "<f4"), ("pz", "<f4"), ("nx", "<f4"), ("ny", "<f4"),
                         ("nz", "<f4"), ("size", "<f4"), ("color", "<f4"), ("r", "<i4"), ("g", "<i4"),
                         ("b", "<i4"), ("weight", "<f4"), ("updateTimes", "<i4"), ("lastUpdate", "<i4")])
SEED_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("normX", "<f4"), ("normY", "<f4"),
                       ("normZ", "<f4"), ("posX", "<f4"), ("posY", "<f4"), ("posZ", "<f4"),
                       ("viewCos", "<f4"), ("meanDepth", "<f4"), ("meanIntensity", "<f4"), ("r", "<i4"),
                       ("g", "<i4"), ("b", "<i4"), ("fused", "u1"), ("stable", "u1"), ("use", "u1"),
                       ("_pad", "u1")])


def build():
    srcs = [os.path.join(ODIR, f) for f in ("orb_oracle.cpp", "surfel_oracle.cpp", "peac_oracle.cpp", "match_oracle.cpp", "Makefile")]
    if not os.path.exists(OLIB) or any(os.path.getmtime(s) > os.path.getmtime(OLIB) for s in srcs if os.path.exists(s)):
        subprocess.check_call(["make", "-C", ODIR, "all", "variants"], stdout=subprocess.DEVNULL)
    return OLIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class Oracle:
    def __init__(self, dll):
        self.dll = dll
        d = dll
        d.mslo_orb_create.restype = C.c_void_p
        d.mslo_orb_create.argtypes = [C.c_int, C.c_float, C.c_int, C.c_int, C.c_int]
        d.mslo_orb_destroy.argtypes = [C.c_void_p]
        d.mslo_orb_tables.argtypes = [C.c_void_p] * 7
        d.mslo_orb_extract.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_void_p, C.c_void_p, C.c_int]
        d.mslo_orb_level_size.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        d.mslo_orb_level.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        d.mslo_orb_candidates.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        d.mslo_resize_linear_u8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int]
        d.mslo_gaussian_blur7.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        d.mslo_gaussian_kernel.argtypes = [C.c_void_p]
        d.mslo_fast_view.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
        d.mslo_distribute_octree.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
        d.mslo_fast_atan2.restype = C.c_float
        d.mslo_fast_atan2.argtypes = [C.c_float, C.c_float]
        d.mslo_sincos.argtypes = [C.c_float, C.c_void_p, C.c_void_p]
        d.mslo_ic_angle.restype = C.c_float
        d.mslo_ic_angle.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        d.mslo_orb_descriptor.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p]

    # ---- ORB ----
    def orb_create(self, nfeatures=1000, scale=1.2, nlevels=8, ini=20, mn=7):
        return OracleOrb(self, nfeatures, scale, nlevels, ini, mn)

    def resize(self, src, dw, dh):
        src = np.ascontiguousarray(src, np.uint8)
        dst = np.zeros((dh, dw), np.uint8)
        self.dll.mslo_resize_linear_u8(_p(src), src.shape[1], src.shape[0], _p(dst), dw, dh)
        return dst

    def blur(self, src):
        src = np.ascontiguousarray(src, np.uint8)
        dst = np.zeros_like(src)
        self.dll.mslo_gaussian_blur7(_p(src), src.shape[1], src.shape[0], _p(dst))
        return dst

    def gaussian_kernel(self):
        k = np.zeros(7, np.int32)
        self.dll.mslo_gaussian_kernel(_p(k))
        return k

    def fast(self, img, threshold):
        img = np.ascontiguousarray(img, np.uint8)
        out = np.zeros((img.size, 3), np.int32)
        n = self.dll.mslo_fast_view(_p(img), img.shape[1], img.shape[0], threshold, _p(out), out.shape[0])
        assert n >= 0
        return out[:n].copy()

    def octree(self, xyr, minX, maxX, minY, maxY, N):
        xyr = np.ascontiguousarray(xyr, np.float32)
        out = np.zeros((max(N + 8, 4 * 8), 3), np.float32)
        n = self.dll.mslo_distribute_octree(_p(xyr), len(xyr), minX, maxX, minY, maxY, N, _p(out), out.shape[0])
        assert n >= 0
        return out[:n].copy()

    def fast_atan2(self, y, x):
        return float(self.dll.mslo_fast_atan2(float(y), float(x)))

    def sincos(self, a):
        s, c = C.c_float(), C.c_float()
        self.dll.mslo_sincos(float(a), C.byref(s), C.byref(c))
        return s.value, c.value

    def ic_angle(self, img, x, y):
        img = np.ascontiguousarray(img, np.uint8)
        return float(self.dll.mslo_ic_angle(_p(img), img.shape[1], img.shape[0], x, y))

    def descriptor(self, blurred, x, y, angle):
        blurred = np.ascontiguousarray(blurred, np.uint8)
        d = np.zeros(32, np.uint8)
        self.dll.mslo_orb_descriptor(_p(blurred), blurred.shape[1], blurred.shape[0], x, y, float(angle), _p(d))
        return d


class OracleOrb:
    def __init__(self, o, nfeatures, scale, nlevels, ini, mn):
        self.o = o
        self.nlevels = nlevels
        self.cap = nfeatures + 2 * nlevels + 64 * nlevels      # wide frames: a level may return its first round's 4 * nIni nodes
        self.h = o.dll.mslo_orb_create(nfeatures, scale, nlevels, ini, mn)

    def __del__(self):
        if getattr(self, "h", None):
            self.o.dll.mslo_orb_destroy(self.h)
            self.h = None

    def tables(self):
        t = [np.zeros(self.nlevels, np.float32) for _ in range(4)]
        per = np.zeros(self.nlevels, np.int32)
        umax = np.zeros(16, np.int32)
        self.o.dll.mslo_orb_tables(self.h, *[_p(a) for a in t], _p(per), _p(umax))
        return t + [per, umax]

    def extract(self, img):
        img = np.ascontiguousarray(img, np.uint8)
        kps = np.zeros(self.cap, KEYPOINT_DTYPE)
        desc = np.zeros((self.cap, 32), np.uint8)
        n = self.o.dll.mslo_orb_extract(self.h, _p(img), img.shape[1], img.shape[0], img.strides[0], _p(kps), _p(desc), self.cap)
        assert n >= 0
        return kps[:n].copy(), desc[:n].copy()

    def level(self, level, blurred=False):
        w, h = C.c_int(), C.c_int()
        self.o.dll.mslo_orb_level_size(self.h, level, C.byref(w), C.byref(h))
        out = np.zeros((h.value, w.value), np.uint8)
        rc = self.o.dll.mslo_orb_level(self.h, level, int(blurred), _p(out))
        assert rc == 0
        return out

    def candidates(self, level, cap=1 << 20):
        out = np.zeros((cap, 3), np.int32)
        n = self.o.dll.mslo_orb_candidates(self.h, level, _p(out), cap)
        assert n >= 0
        return out[:n].copy()


_cached = None


def load():
    global _cached
    if _cached is None:
        _cached = Oracle(C.CDLL(build()))
    return _cached


# ------------------------------------------------------------------------------------------------
# Surfel fusion
# ------------------------------------------------------------------------------------------------
class OracleSurfel:
    def __init__(self, w, h, fx, fy, cx, cy, far=30.0, near=0.5):
        self.o = load()
        d = self.o.dll
        d.mslo_sf_create.restype = C.c_void_p
        d.mslo_sf_create.argtypes = [C.c_int, C.c_int] + [C.c_float] * 6
        d.mslo_sf_destroy.argtypes = [C.c_void_p]
        d.mslo_sf_fuse.restype = C.c_long
        d.mslo_sf_fuse.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                   C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        d.mslo_sf_map_set.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        d.mslo_sf_map_size.restype = C.c_size_t
        d.mslo_sf_map_size.argtypes = [C.c_void_p]
        d.mslo_sf_map_get.argtypes = [C.c_void_p, C.c_void_p]
        d.mslo_sf_fuse_map.restype = C.c_long
        d.mslo_sf_fuse_map.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                       C.c_void_p]
        d.mslo_sf_seeds.argtypes = [C.c_void_p, C.c_void_p]
        d.mslo_sf_set_threads.argtypes = [C.c_void_p, C.c_int]
        d.mslo_sf_index.argtypes = [C.c_void_p, C.c_void_p]
        d.mslo_fuse_map_compact.restype = C.c_size_t
        d.mslo_fuse_map_compact.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        self.w, self.h = w, h
        self.hd = d.mslo_sf_create(w, h, fx, fy, cx, cy, far, near)

    def __del__(self):
        if getattr(self, "hd", None):
            self.o.dll.mslo_sf_destroy(self.hd)
            self.hd = None

    def fuse(self, ref, gray, depth, member, pose, local):
        """SurfelFusion::fuseInitializeMap: returns (updated local copy, new surfels)."""
        local = local.copy()
        new = np.zeros((self.w // 8) * (self.h // 8), SURFEL_DTYPE)
        n = self.o.dll.mslo_sf_fuse(self.hd, ref, _p(gray), gray.strides[0], _p(depth), depth.strides[0], _p(member),
                                    member.strides[0], _p(pose), _p(local), len(local), _p(new), len(new))
        assert n >= 0
        return local, new[:n].copy()

    def set_threads(self, threaded):
        """True: the reference's THREAD_NUM = 10 std::threads per stage (CPU timing baseline only; racy like the reference)."""
        self.o.dll.mslo_sf_set_threads(self.hd, 1 if threaded else 0)

    def map_set(self, m):
        m = np.ascontiguousarray(m)
        self.o.dll.mslo_sf_map_set(self.hd, _p(m), len(m))

    def map_get(self):
        out = np.zeros(self.o.dll.mslo_sf_map_size(self.hd), SURFEL_DTYPE)
        self.o.dll.mslo_sf_map_get(self.hd, _p(out))
        return out

    def fuse_map(self, ref, gray, depth, member, pose):
        return int(self.o.dll.mslo_sf_fuse_map(self.hd, ref, _p(gray), gray.strides[0], _p(depth), depth.strides[0], _p(member),
                                               member.strides[0], _p(pose)))

    def seeds(self):
        out = np.zeros((self.w // 8) * (self.h // 8), SEED_DTYPE)
        self.o.dll.mslo_sf_seeds(self.hd, _p(out))
        return out

    def index(self):
        out = np.zeros((self.h, self.w), np.int32)
        self.o.dll.mslo_sf_index(self.hd, _p(out))
        return out


def fuse_map_compact(local, new):
    buf = np.zeros(len(local) + len(new), SURFEL_DTYPE)
    buf[:len(local)] = local
    new = np.ascontiguousarray(new)
    n = load().dll.mslo_fuse_map_compact(_p(buf), len(local), _p(new), len(new))
    return buf[:n].copy()


# ------------------------------------------------------------------------------------------------
# Frame post-ORB steps
# ------------------------------------------------------------------------------------------------
FRAME_PARAMS_DTYPE = np.dtype([(n, "<f4") for n in ("fx", "fy", "cx", "cy", "k1", "k2", "p1", "p2", "k3", "bf", "minX", "maxX", "minY", "maxY")])


def frame_params(fx, fy, cx, cy, bf, width, height, k1=0.0, k2=0.0, p1=0.0, p2=0.0, k3=0.0):
    p = np.zeros(1, FRAME_PARAMS_DTYPE)
    for k, v in dict(fx=fx, fy=fy, cx=cx, cy=cy, k1=k1, k2=k2, p1=p1, p2=p2, k3=k3, bf=bf).items():
        p[k] = v
    d = load().dll
    d.mslo_frame_image_bounds.argtypes = [C.c_void_p, C.c_int, C.c_int]
    d.mslo_frame_image_bounds(_p(p), width, height)
    return p


def frame_epilogue(params, kps, depth):
    d = load().dll
    d.mslo_frame_epilogue.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t] + [C.c_void_p] * 4
    n = len(kps)
    kps = np.ascontiguousarray(kps); depth = np.ascontiguousarray(depth, np.float32)
    un = np.zeros((n, 2), np.float32); dep = np.zeros(n, np.float32); ur = np.zeros(n, np.float32); cell = np.zeros(n, np.int32)
    d.mslo_frame_epilogue(_p(params), _p(kps), n, _p(depth), depth.strides[0], _p(un), _p(dep), _p(ur), _p(cell))
    return un, dep, ur, cell


def peac_block_stats(depth_u16, fx, fy, cx, cy, depth_map_factor, window=(10, 10), depth_alpha=0.04, depth_change_tol=0.02, init_loose=False):
    """oracle/peac_oracle.cpp on one [H, W] uint16 depth image -> (cloud [ch*cw, 3], stats)."""
    from manhattanslam_amd import PEAC_STATS_DTYPE
    d = np.ascontiguousarray(depth_u16, np.uint16)
    H, W = d.shape
    cw, ch = (W + 1) // 2, (H + 1) // 2
    stats = np.zeros((cw // window[0]) * (ch // window[1]), PEAC_STATS_DTYPE)
    cloud = np.zeros((cw * ch, 3), np.float64)
    f = load().dll.mslo_peac_block_stats
    f.restype = None
    f.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int,
                  C.c_double, C.c_double, C.c_int, C.c_void_p, C.c_void_p]
    f(_p(d), d.strides[0], W, H, fx, fy, cx, cy, depth_map_factor, window[0], window[1], depth_alpha, depth_change_tol, 1 if init_loose else 0,
      _p(cloud), _p(stats))
    return cloud, stats


# ------------------------------------------------------------------------------------------------
# Hamming matching by projection
# ------------------------------------------------------------------------------------------------
MATCH_PARAMS_DTYPE = np.dtype([(n, "<f4") for n in ("fx", "fy", "cx", "cy", "bf", "minX", "maxX", "minY", "maxY", "th")] +
                              [("check_orientation", "<i4"), ("nlevels", "<i4"), ("scale_factors", "<f4", (16,))])


def search_by_projection(params, cur, last, Tcw_cur, Tcw_last):
    """oracle/match_oracle.cpp for ONE pair: same dict layout as manhattanslam_amd.match.search_by_projection_batch."""
    f = load().dll.mslo_search_by_projection
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 5 + [C.c_int] + [C.c_void_p] * 8
    c = {k: np.ascontiguousarray(v) for k, v in cur.items()}
    l = {k: np.ascontiguousarray(v) for k, v in last.items()}
    tc = np.ascontiguousarray(np.asarray(Tcw_cur, np.float32)[:3, :4]); tl = np.ascontiguousarray(np.asarray(Tcw_last, np.float32)[:3, :4])
    n = len(c["kps"])
    out = np.zeros(max(n, 1), np.int32)
    nm = f(_p(params), n, _p(c["kps"]), _p(c["un_xy"].astype(np.float32)), _p(c["uright"].astype(np.float32)), _p(c["grid_cell"].astype(np.int32)),
           _p(c["desc"]), len(l["xyz"]), _p(l["xyz"].astype(np.float32)), _p(l["desc"]), _p(l["flags"].astype(np.uint8)),
           _p(l["octave"].astype(np.int32)), _p(l["angle"].astype(np.float32)), _p(tc), _p(tl), _p(out))
    return out[:n].copy(), nm


def descriptor_distance(a, b):
    f = load().dll.mslo_descriptor_distance
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_void_p]
    a = np.ascontiguousarray(a, np.uint8).reshape(-1, 32); b = np.ascontiguousarray(b, np.uint8).reshape(-1, 32)
    return np.array([f(_p(a[i]), _p(b[i])) for i in range(len(a))], np.int32)


PEAC_PARAMS_DTYPE = np.dtype([(n, "<i4") for n in ("window_w", "window_h", "min_support", "max_step", "do_refine", "erode_type", "init_loose", "_pad")] +
                             [(n, "<f8") for n in ("depth_sigma", "std_tol_init", "std_tol_merge", "z_near", "z_far", "angle_near", "angle_far",
                                                   "similarity_th_merge", "similarity_th_refine", "depth_alpha", "depth_change_tol")])
PEAC_BLOCK_DTYPE = np.dtype([(n, "<f8") for n in ("sx", "sy", "sz", "sxx", "syy", "szz", "sxy", "syz", "sxz")] + [("N", "<i4"), ("nouse", "<i4")] +
                            [("center", "<f8", (3,)), ("normal", "<f8", (3,)), ("mse", "<f8"), ("curvature", "<f8")])


def peac_default_params():
    p = np.zeros(1, PEAC_PARAMS_DTYPE)
    load().dll.mslo_peac_default_params(_p(p))
    return p


def peac_run(depth_u16, fx, fy, cx, cy, depth_map_factor, params=None):
    """oracle/peac_oracle.cpp: the whole plane extractor on one [H, W] uint16 depth image -> (membership [ch, cw] i32, n_planes, blocks)."""
    d = np.ascontiguousarray(depth_u16, np.uint16)
    H, W = d.shape
    cw, ch = (W + 1) // 2, (H + 1) // 2
    prm = peac_default_params() if params is None else params
    nb = (cw // int(prm["window_w"][0])) * (ch // int(prm["window_h"][0]))
    member = np.zeros((ch, cw), np.int32)
    blocks = np.zeros(nb, PEAC_BLOCK_DTYPE)
    n = C.c_int32(0)
    f = load().dll.mslo_peac_run
    f.restype = None
    f.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p,
                  C.c_void_p]
    f(_p(d), d.strides[0], W, H, fx, fy, cx, cy, depth_map_factor, _p(prm), _p(member), C.byref(n), _p(blocks))
    return member, n.value, blocks


PEAC_PLANE_DTYPE = np.dtype([("normal", "<f8", (3,)), ("center", "<f8", (3,)), ("mse", "<f8"), ("N", "<i4"), ("_pad", "<i4")])


def peac_last_planes(n_vertices):
    """extractedPlanes + plane_vertices_ of this thread's last peac_run: (planes PEAC_PLANE_DTYPE [n], list of int32 vertex-index arrays)."""
    planes = np.zeros(1024, PEAC_PLANE_DTYPE); offsets = np.zeros(1025, np.int32); indices = np.zeros(n_vertices, np.int32)
    f = load().dll.mslo_peac_last_planes
    f.restype = C.c_int32
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    k = f(_p(planes), _p(offsets), _p(indices))
    return planes[:k].copy(), [indices[offsets[j]:offsets[j + 1]].copy() for j in range(k)]


def eig33sym(K):
    K = np.ascontiguousarray(K, np.float64)
    s = np.zeros(3); V = np.zeros((3, 3))
    f = load().dll.mslo_eig33sym
    f.restype = None
    f.argtypes = [C.c_void_p] * 3
    f(_p(K), _p(s), _p(V))
    return s, V
