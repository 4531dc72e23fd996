"""CPU tests of the surfel-fusion oracle: closed-form known answers + the committed golden vector."""
import hashlib
import os

import numpy as np
import pytest

from manhattanslam_amd import synth
from tests.oracle_lib import OracleSurfel, SURFEL_DTYPE, fuse_map_compact, load

GOLD = os.path.join(os.path.dirname(__file__), "golden")
I = synth.TUM1
IDENT = np.eye(4, dtype=np.float32).T.reshape(16).copy()


def _mk(w=640, h=480, intr=I):
    return OracleSurfel(w, h, intr["fx"], intr["fy"], intr["cx"], intr["cy"], 30.0, 0.5)


def _plane_depth(w, h, n, d, intr=I):
    """z-depth of the plane n.X + d = 0 seen by the camera at the origin."""
    u, v = np.meshgrid(np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64))
    rx, ry = (u - intr["cx"]) / intr["fx"], (v - intr["cy"]) / intr["fy"]
    return (-d / (n[0] * rx + n[1] * ry + n[2])).astype(np.float32)


def test_planar_scene_gives_plane_normal_and_positions():
    n = np.array([0.2, -0.1, -1.0]); n /= np.linalg.norm(n)
    depth = _plane_depth(640, 480, n, 2.0)
    gray = np.full((480, 640), 120, np.uint8)
    member = np.full((240, 320), -1, np.int32)
    sf = _mk()
    local = np.zeros(0, SURFEL_DTYPE)
    _, new = sf.fuse(0, gray, depth, member, IDENT, local)
    s = sf.seeds()
    inner = s.reshape(60, 80)[2:-2, 2:-2].ravel()
    nn = np.stack([inner["normX"], inner["normY"], inner["normZ"]], 1)
    assert np.abs(nn - n[None, :]).max() < 2e-3                     # robust plane fit recovers the plane normal
    pos = np.stack([inner["posX"], inner["posY"], inner["posZ"]], 1).astype(np.float64)
    assert np.abs(pos @ n + 2.0).max() < 2e-3                       # seed centres lie on the plane
    assert np.all(inner["viewCos"] > 0.1) and np.all(inner["use"] == 1)
    assert len(new) >= inner.size                                   # nothing to fuse with: every valid seed spawns a surfel
    assert np.all(new["updateTimes"] == 1) and np.all(new["lastUpdate"] == 0)
    w = np.minimum(1.0 / new["pz"].astype(np.float64) ** 2, 1.0)
    assert np.abs(new["weight"] - w).max() < 1e-3                   # getWeight = min(1/z^2, 1) (identity pose: z = pz)


def test_single_surfel_fuses_by_weighted_mean_and_deletion_rules():
    depth = np.full((480, 640), 2.0, np.float32)
    gray = np.full((480, 640), 50, np.uint8)
    member = np.full((240, 320), -1, np.int32)
    sf = _mk()
    loc = np.zeros(5, SURFEL_DTYPE)
    # 0: on the wall, facing the camera -> fused;  1: 1.5 m in front of the wall -> occlusion delete
    # 2: normal facing away -> delete;  3: stale and rarely updated -> delete;  4: behind fuseNear -> untouched
    loc["px"] = [0.1, 0.1, -0.2, 0.3, 0.0]; loc["py"] = [0.05, 0.05, 0.1, 0.0, 0.0]; loc["pz"] = [2.03, 0.5 + 1e-3, 2.0, 2.0, 0.2]
    loc["nz"] = [-1, -1, 1, -1, -1]
    loc["size"] = 1.0; loc["weight"] = [3.0, 1, 1, 1, 1]; loc["updateTimes"] = [7, 7, 7, 2, 7]; loc["lastUpdate"] = [9, 9, 9, 1, 9]
    out, new = sf.fuse(10, gray, depth, member, IDENT, loc)
    assert out["updateTimes"].tolist() == [8, 0, 0, 0, 7]
    assert out["lastUpdate"].tolist() == [10, 9, 9, 1, 9]
    wn = min(1 / 2.0 ** 2, 1.0)
    assert abs(out["pz"][0] - (2.03 * 3.0 + wn * 2.0) / (3.0 + wn)) < 1e-4      # weighted mean with w_new = min(1/z^2, 1)
    assert abs(out["weight"][0] - (3.0 + wn)) < 1e-6
    assert abs(out["nz"][0] + 1) < 1e-4 and out["color"][0] == 50 and out["size"][0] < 1.0
    assert out[4].tobytes() == loc[4].tobytes()


def test_membership_masks_seeds():
    gray, depth, member, pose = synth.surfel_frame(0, variant="B")
    sf = _mk()
    sf.fuse(0, gray, depth, member, pose, np.zeros(0, SURFEL_DTYPE))
    s = sf.seeds().reshape(60, 80)
    idx = sf.index()
    ys, xs = np.nonzero(np.repeat(np.repeat(member, 2, 0), 2, 1) != -1)
    assert np.all(idx[ys, xs] == 0)                               # plane pixels are never assigned
    cy, cx = np.arange(60) * 8 + 4, np.arange(80) * 8 + 4
    use = member[np.ix_(cy // 2, cx // 2)] == -1
    assert np.array_equal(s["use"].astype(bool), use)


def test_compaction_rule_matches_prefix_sum_model():
    """Back-to-front refill + tail compaction == 'new k -> k-th largest hole; the a-th smallest leftover hole receives
    resolve(nFinal + a)', resolve following relay holes inside the tail (the formulation the GPU kernels use)."""
    rng = np.random.default_rng(0)
    cases = [(0, 0, 3), (10, 1.0, 3), (10, 1.0, 12), (1000, 0.3, 50), (1000, 0.02, 50), (1000, 0.9, 5), (257, 0.5, 0),
             (300, 1.0, 1), (300, 0.97, 7)]
    for n, pdel, k in cases:
        loc = np.zeros(n, SURFEL_DTYPE)
        loc["px"] = np.arange(n); loc["updateTimes"] = np.where(rng.random(n) < pdel, 0, 1)
        new = np.zeros(k, SURFEL_DTYPE)
        new["px"] = 10000 + np.arange(k); new["updateTimes"] = 1
        got = fuse_map_compact(loc, new)
        holes = np.flatnonzero(loc["updateTimes"] == 0)
        D = len(holes)
        model = list(loc["px"])
        for j in range(min(k, D)):
            model[holes[D - 1 - j]] = 10000 + j
        model += [10000 + j for j in range(D, k)]
        if D > k:
            left = holes[:D - k].tolist()
            R = len(left)
            nfin = n - R
            pos = {h: i for i, h in enumerate(left)}
            final = list(model)
            for a, hh in enumerate(h for h in left if h < nfin):
                p = nfin + a
                while p in pos:
                    p = n - (R - pos[p])
                final[hh] = model[p]
            model = final[:nfin]
        assert got["px"].tolist() == [float(v) for v in model], (n, pdel, k)


def test_inverse4_matches_numpy():
    dll = load().dll
    import ctypes as C
    rng = np.random.default_rng(4)
    for k in range(5):
        pose = synth.camera_pose(k * 17)
        inv = np.zeros(16, np.float32)
        dll.mslo_inverse4f(pose.ctypes.data_as(C.c_void_p), inv.ctypes.data_as(C.c_void_p))
        ref = np.linalg.inv(pose.reshape(4, 4).T.astype(np.float64))
        assert np.abs(inv.reshape(4, 4).T - ref).max() < 1e-6


def _same_records(a, b):
    """Bit-identical structured arrays, except that a NaN equals a NaN whatever its sign / payload: which of the two operands' NaNs an x86 instruction
    hands on (and the sign of a freshly made one) follows the compiler's operand order, which changes with unrelated edits of the oracle; the
    reference's own NaNs are no more defined than that."""
    a, b = np.asarray(a), np.asarray(b)
    if a.dtype != b.dtype or a.shape != b.shape:
        return False
    for name in a.dtype.names:
        x, y = a[name], b[name]
        if x.dtype.kind == "f":
            assert x.dtype.itemsize == 4, name                     # (the bit view below is for binary32 fields)
            # byte equality everywhere except where the GOLDEN value itself is a NaN: there any NaN will do (never a finite value, and a NaN
            # where the golden vector holds a number is a mismatch)
            same = (x.view(np.uint32) == y.view(np.uint32)) | (np.isnan(y) & np.isnan(x))
        else:
            same = x == y
        if not np.all(same):
            return False
    return True



def test_golden_surfel():
    g = np.load(os.path.join(GOLD, "surfel_640x480_B.npz"))
    k = int(g["frame"])
    gray, depth, member, pose = synth.surfel_frame(k, variant="B")
    assert hashlib.sha256(depth.tobytes()).hexdigest() == str(g["depth_sha256"])
    local = synth.surfel_map(int(g["n_local"]), ref=k).astype(SURFEL_DTYPE)
    sf = _mk()
    lo, no = sf.fuse(k, gray, depth, member, pose, local)
    assert _same_records(no, g["new_surfels"])
    assert _same_records(lo[g["changed_index"]], g["changed_surfels"])
    assert _same_records(sf.seeds(), g["seeds"])
    assert hashlib.sha256(sf.index().tobytes()).hexdigest() == str(g["index_sha256"])


def test_fabs_pin_known_answer():
    """src/SurfelFusion.cpp:488 sums three unqualified fabs() of floats.  The oracle pins the float overload (a float chain);
    C's ::fabs(double) would add in double and round once.  Known answer where the two readings differ by one ulp."""
    import ctypes as C
    from tests import oracle_lib
    f = oracle_lib.load().dll.mslo_update_diff
    f.restype = C.c_float
    f.argtypes = [C.c_float] * 6 + [C.c_int]
    f32 = np.float32
    # |d1| + |d2| + |d3| with d = (0.1, 0.05, 0.05000001): float chain rounds after every add
    rng = np.random.default_rng(5)
    found = None
    for _ in range(20000):
        a, b, c = (f32(x) for x in rng.uniform(0.01, 0.2, 3))
        chain = f32(f32(a + b) + c)
        once = f32(np.float64(a) + np.float64(b) + np.float64(c))
        if chain != once:
            found = (a, b, c, chain, once)
            break
    assert found is not None
    a, b, c, chain, once = found
    z = f32(0)
    assert f32(f(float(a), 0.0, float(b), 0.0, float(c), 0.0, 0)) == chain        # pinned: std::fabs(float)
    assert f32(f(float(a), 0.0, float(b), 0.0, float(c), 0.0, 1)) == once         # alternative: ::fabs(double)
    assert abs(float(chain) - float(once)) <= float(np.spacing(chain)) and z == 0


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_used_seed_always_owns_its_centre_pixel(seed):
    """The chunk-abort `return` of src/SurfelFusion.cpp:473-474 needs a used seed that owns no pixel.  It cannot happen: the pixel at the lattice centre of a used seed is free and has that seed as its only candidate (x mod 8 == 4), so
    it is assigned in pass 0 and never moves.  Checked on adversarial inputs: white noise, stripes that pull every other pixel away, NaN /
    zero / huge depth, random membership holes."""
    rng = np.random.default_rng(seed)
    w, h = (160, 120) if seed != 3 else (163, 125)      # (a size that is not a multiple of 8 as well)
    gray = rng.integers(0, 256, (h, w)).astype(np.uint8)
    if seed == 1:
        gray[:, ::2] = 0; gray[:, 1::2] = 255
    if seed == 2:
        gray[:] = np.where((np.arange(w)[None, :] // 4 + np.arange(h)[:, None] // 4) & 1, 255, 0)
        gray[4::8, 4::8] = 128                                    # centre pixels unlike everything around them
    depth = rng.uniform(0.3, 5.0, (h, w)).astype(np.float32)
    depth[rng.random((h, w)) < 0.2] = 0.0
    depth[rng.random((h, w)) < 0.02] = np.nan
    depth[rng.random((h, w)) < 0.02] = 1e30
    member = np.full(((h + 1) // 2, (w + 1) // 2), -1, np.int32)
    member[rng.random(member.shape) < 0.15] = 3
    sf = OracleSurfel(w, h, 130.0, 130.0, 80.0, 60.0, 30.0, 0.5)
    pose = np.eye(4, dtype=np.float32).T.reshape(16).copy()
    sf.fuse(0, gray, depth, member, pose, np.zeros(0, SURFEL_DTYPE))
    seeds, index = sf.seeds(), sf.index()
    used = np.flatnonzero(seeds["use"])
    assert len(used) > 100
    cy, cx = (used // (w // 8)) * 8 + 4, (used % (w // 8)) * 8 + 4
    assert np.array_equal(index[cy, cx], used)


def test_golden_clutter():
    """Committed golden vector of the furnished room (round 4): seeds, index map, changed / new surfels of SurfelFusion on a mostly-in-view map,
    and the plane extractor's membership image on the same depth frame."""
    from tests import oracle_lib
    g = np.load(os.path.join(GOLD, "clutter_640x480.npz"))
    k = int(g["frame"])
    sc = synth.clutter_scene()
    gray, depth, member, pose, _ = synth.clutter_frame(k, scene=sc)
    assert hashlib.sha256(depth.tobytes()).hexdigest() == str(g["depth_sha256"]) and hashlib.sha256(gray.tobytes()).hexdigest() == str(g["gray_sha256"])
    local = synth.surfel_map_dense(int(g["n_local"]), ref=k, scene=sc, k_lo=k - 25, k_hi=k + 35, flip=0.05, floating=0.02, min_update_times=1).astype(SURFEL_DTYPE)
    assert hashlib.sha256(local.tobytes()).hexdigest() == str(g["map_sha256"])
    sf = _mk()
    lo, no = sf.fuse(k, gray, depth, member, pose, local)
    assert _same_records(no, g["new_surfels"])
    assert _same_records(lo[g["changed_index"]], g["changed_surfels"])
    assert _same_records(sf.seeds(), g["seeds"])
    assert hashlib.sha256(sf.index().tobytes()).hexdigest() == str(g["index_sha256"])
    I = synth.TUM1
    mem, npl, _ = oracle_lib.peac_run(synth.depth_u16(depth), I["fx"], I["fy"], I["cx"], I["cy"], np.float32(1.0 / 5000.0))
    assert npl == int(g["peac_nplanes"]) and np.array_equal(mem, g["peac_membership"].astype(np.int32))
