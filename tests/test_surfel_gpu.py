"""GPU parity: HIP surfel fusion (through the C ABI) vs the CPU oracle.

Tolerance (BASELINE.json north_star): positions / normals / radii within 1e-4 of the CPU path; integer fields
and array order identical.  In practice everything except the FP64 plane-fit reductions is bit-identical."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 1e-4
FLOAT_FIELDS = ("px", "py", "pz", "nx", "ny", "nz", "size", "color", "weight")
INT_FIELDS = ("r", "g", "b", "updateTimes", "lastUpdate")


def assert_surfels_close(a, b, what=""):
    assert len(a) == len(b), (what, len(a), len(b))
    for f in INT_FIELDS:
        assert np.array_equal(a[f], b[f]), (what, f, np.flatnonzero(a[f] != b[f])[:10])
    for f in FLOAT_FIELDS:
        d = np.abs(a[f].astype(np.float64) - b[f].astype(np.float64))
        assert np.all(np.isnan(a[f]) == np.isnan(b[f])), (what, f)
        d = d[~np.isnan(d)]
        assert d.size == 0 or d.max() <= TOL, (what, f, d.max())


def assert_seeds_close(a, b):
    for f in ("r", "g", "b", "fused", "stable", "use"):
        assert np.array_equal(a[f], b[f]), (f, np.flatnonzero(a[f] != b[f])[:10])
    for f in ("x", "y", "meanIntensity"):
        assert np.array_equal(a[f].view(np.int32), b[f].view(np.int32)), f
    for f in ("size", "normX", "normY", "normZ", "posX", "posY", "posZ", "viewCos", "meanDepth"):
        d = np.abs(a[f].astype(np.float64) - b[f].astype(np.float64))
        assert np.all(np.isnan(a[f]) == np.isnan(b[f])), f
        d = d[~np.isnan(d)]
        assert d.max() <= TOL, (f, d.max(), np.argmax(d))


def _mk(intr, w=640, h=480):
    from manhattanslam_amd import SurfelFusion
    from tests.oracle_lib import OracleSurfel
    g = SurfelFusion(w, h, intr["fx"], intr["fy"], intr["cx"], intr["cy"], 30.0, 0.5)
    o = OracleSurfel(w, h, intr["fx"], intr["fy"], intr["cx"], intr["cy"], 30.0, 0.5)
    return g, o


@pytest.mark.parametrize("variant", ["A", "B"])
def test_host_vector_mode_matches_oracle(oracle, variant):
    """SurfelFusion::fuseInitializeMap drop-in: local vector updated in place + new surfel list."""
    from manhattanslam_amd import synth, SURFEL_DTYPE
    g, o = _mk(synth.TUM1)
    local = synth.surfel_map(50000, ref=3).astype(SURFEL_DTYPE)
    gray, depth, member, pose = synth.surfel_frame(3, variant=variant)
    lo, no = o.fuse(3, gray, depth, member, pose, local)
    lg = local.copy()
    ng = g.fuseInitializeMap(3, gray, depth, member, pose, lg)
    assert np.array_equal(g.debug_index(), o.index())
    assert_seeds_close(g.debug_seeds(), o.seeds())
    assert_surfels_close(lg, lo, "local")
    assert_surfels_close(ng, no, "new")
    assert (lg["updateTimes"] == 0).sum() > 100 and (lg["lastUpdate"] == 3).sum() > 1000 and len(ng) > 50
    g.close()


def test_resident_sequence_matches_oracle(oracle):
    """Five keyframes on the resident map incl. slot refill and tail compaction (fuseMap)."""
    from manhattanslam_amd import synth, SURFEL_DTYPE
    g, o = _mk(synth.TUM1)
    m = synth.surfel_map(120000, ref=0).astype(SURFEL_DTYPE)
    g.map_reserve(300000)
    g.map_upload(m)
    o.map_set(m)
    for k in range(5):
        gray, depth, member, pose = synth.surfel_frame(k, variant="B" if k == 2 else "A")
        g.fuse_resident(k, gray, depth, member, pose)
        n_new = o.fuse_map(k, gray, depth, member, pose)
        c = g.counters()
        assert c["n_new"] == n_new
        mg, mo = g.map_download(), o.map_get()
        assert c["n_live_after"] == len(mo)
        assert_surfels_close(mg, mo, f"map after keyframe {k}")
    g.close()


def test_batched_keyframes_match_sequential_oracle(oracle):
    """fuse_resident_batch == the same keyframes one after another (two overlapping batches, device + host input)."""
    import torch
    from manhattanslam_amd import synth, SURFEL_DTYPE
    g, o = _mk(synth.TUM1)
    g.set_batch_capacity(4)
    m = synth.surfel_map(150000, ref=0).astype(SURFEL_DTYPE)
    g.map_reserve(400000)
    g.map_upload(m)
    o.map_set(m)
    frames = [synth.surfel_frame(k, variant="B" if k % 3 == 1 else "A") for k in range(7)]
    for k, (gray, depth, member, pose) in enumerate(frames):
        o.fuse_map(k, gray, depth, member, pose)
    grays = np.stack([f[0] for f in frames]); depths = np.stack([f[1] for f in frames]); members = np.stack([f[2] for f in frames])
    poses = [f[3] for f in frames]
    # batch 1: host memory, 4 keyframes; batch 2: device memory, 3 keyframes, issued without waiting for batch 1
    g.fuse_resident_batch([0, 1, 2, 3], grays[:4], depths[:4], members[:4], poses[:4])
    dg, dd, dm = (torch.from_numpy(a[4:]).cuda() for a in (grays, depths, members))
    g.fuse_resident_batch([4, 5, 6], dg, dd, dm, poses[4:], device=True)
    mg, mo = g.map_download(), o.map_get()
    assert_surfels_close(mg, mo, "map after 7 batched keyframes")
    assert np.array_equal(g.debug_index(), o.index())
    assert_seeds_close(g.debug_seeds(), o.seeds())
    g.close()


def test_strided_images(oracle):
    """Row strides larger than the width (cv::Mat ROI style) incl. the Vec3b flat-offset quirk."""
    from manhattanslam_amd import synth, SURFEL_DTYPE
    g, o = _mk(synth.TUM1)
    gray, depth, member, pose = synth.surfel_frame(1)
    G = np.zeros((480, 700), np.uint8); G[:, :640] = gray; G[:, 640:] = 17
    Dp = np.zeros((480, 648), np.float32); Dp[:, :640] = depth
    Mb = np.full((240, 330), 5, np.int32); Mb[:, :320] = member
    gv, dv, mv = G[:, :640], Dp[:, :640], Mb[:, :320]
    local = synth.surfel_map(20000, ref=1).astype(SURFEL_DTYPE)
    lo, no = o.fuse(1, gv, dv, mv, pose, local)
    lg = local.copy()
    ng = g.fuseInitializeMap(1, gv, dv, mv, pose, lg)
    assert_seeds_close(g.debug_seeds(), o.seeds())
    assert_surfels_close(lg, lo)
    assert_surfels_close(ng, no)
    g.close()


def test_rows_spanning_4gb_are_refused():
    """The kernels address an image with 32-bit byte offsets (round 6): a row stride that makes the rows of ONE image span 4 GB or more is refused
    with MSL_ERR_INVALID before anything is launched (include/msl.h); nothing is read through the pointers."""
    import ctypes as C
    import torch
    from manhattanslam_amd import SurfelFusion, synth
    from manhattanslam_amd._lib import lib, MSL_MEM_DEVICE
    I = synth.TUM1
    sf = SurfelFusion(640, 480, I["fx"], I["fy"], I["cx"], I["cy"], 30.0, 0.5)
    sf.set_batch_capacity(1)
    g = torch.zeros((480, 640), dtype=torch.uint8, device="cuda"); d = torch.zeros((480, 640), dtype=torch.float32, device="cuda")
    m = torch.zeros((240, 320), dtype=torch.int32, device="cuda")
    refs = np.zeros(1, np.int32); pose = np.eye(4, dtype=np.float32).T.reshape(-1).copy()
    def call(gs, ds, ms):
        return lib.msl_sf_fuse_resident_batch(sf._h, 1, refs.ctypes.data_as(C.c_void_p), C.c_void_p(g.data_ptr()), C.c_size_t(gs), C.c_size_t(0),
                                              C.c_void_p(d.data_ptr()), C.c_size_t(ds), C.c_size_t(0), C.c_void_p(m.data_ptr()), C.c_size_t(ms), C.c_size_t(0),
                                              MSL_MEM_DEVICE, pose.ctypes.data_as(C.c_void_p))
    big = (1 << 32) // 480 + 4
    assert call(big, 4 * 640, 4 * 320) != 0          # gray rows span >= 4 GB
    assert call(640, big & ~3, 4 * 320) != 0         # depth rows
    assert call(640, 4 * 640, ((1 << 32) // 240 + 8) & ~3) != 0   # membership rows
    assert call(640, 4 * 640, 4 * 320) == 0          # the same images with their real strides
    sf.sync(); sf.close()


def test_icl_negative_fy_and_empty_map(oracle):
    from manhattanslam_amd import synth, SURFEL_DTYPE
    g, o = _mk(synth.ICL)
    gray, depth, member, pose = synth.surfel_frame(1, intr=synth.ICL)
    local = np.zeros(0, SURFEL_DTYPE)
    lo, no = o.fuse(1, gray, depth, member, pose, local)
    ng = g.fuseInitializeMap(1, gray, depth, member, pose, local)
    assert_seeds_close(g.debug_seeds(), o.seeds())
    assert_surfels_close(ng, no, "new")
    assert len(ng) > 1000
    g.close()


def test_degenerate_depth(oracle):
    """All-invalid depth: no seed gets a plane, nothing fuses, no new surfels; then a depth image with holes."""
    from manhattanslam_amd import synth, SURFEL_DTYPE
    g, o = _mk(synth.TUM1)
    gray, depth, member, pose = synth.surfel_frame(0)
    zero = np.zeros_like(depth)
    local = synth.surfel_map(2000, ref=0).astype(SURFEL_DTYPE)
    lo, no = o.fuse(0, gray, zero, member, pose, local)
    lg = local.copy()
    ng = g.fuseInitializeMap(0, gray, zero, member, pose, lg)
    assert len(ng) == len(no) == 0
    assert_surfels_close(lg, lo)
    holes = depth.copy()
    holes[100:300, 200:500] = 0
    holes[::7, ::5] = 0
    lo, no = o.fuse(1, gray, holes, member, pose, local)
    lg = local.copy()
    ng = g.fuseInitializeMap(1, gray, holes, member, pose, lg)
    assert np.array_equal(g.debug_index(), o.index())
    assert_seeds_close(g.debug_seeds(), o.seeds())
    assert_surfels_close(lg, lo)
    assert_surfels_close(ng, no)
    g.close()


def test_stable_seeds_and_relaxation_chains(oracle):
    """Flat grey regions next to textured ones: more than half of the seeds turn stable after the first update, so the pixel passes skip most
    pixels, processed pixels un-stabilise neighbouring seeds in raster order (the `stable` chains the relaxation restates), and the pixel pass's
    candidate records carry stable flags.  Index map and seeds as the oracle's, two keyframes in a row on the same handle."""
    from manhattanslam_amd import synth, SURFEL_DTYPE
    g, o = _mk(synth.TUM1)
    local = synth.surfel_map(3000, ref=5).astype(SURFEL_DTYPE)
    lg = local.copy()
    for k in (5, 6):
        gray, depth, member, pose = synth.surfel_frame(k)
        g2 = np.full_like(gray, 120)
        g2[:, 200:260] = gray[:, 200:260]
        g2[300:, 400:] = 60
        g2[50:120, 420:600] = np.random.default_rng(k).integers(0, 255, (70, 180), dtype=np.uint8)
        lo, no = o.fuse(k, g2, depth, member, pose, local)
        ng = g.fuseInitializeMap(k, g2, depth, member, pose, lg)
        so = o.seeds()
        assert int((so["stable"] & so["use"]).sum()) > 2000
        assert np.array_equal(g.debug_index(), o.index())
        assert_seeds_close(g.debug_seeds(), so)
        assert_surfels_close(lg, lo, "local")
        assert_surfels_close(ng, no, "new")
        local = lo
    g.close()


def test_compaction_matches_literal_loop(oracle):
    """Slot refill / tail compaction against the literal back-to-front loop for adversarial delete patterns."""
    from manhattanslam_amd import synth, SURFEL_DTYPE
    from tests.oracle_lib import fuse_map_compact
    g, o = _mk(synth.TUM1)
    rng = np.random.default_rng(3)
    gray, depth, member, pose = synth.surfel_frame(0)
    for trial, (n, pdel) in enumerate([(5000, 0.0), (5000, 0.9), (20000, 0.5), (9000, 1.0), (4097, 0.3), (30000, 1.0), (30000, 0.95),
                                       (3000, 0.05), (3000, 0.12), (2500000, 0.0005),
                                       (400000, 1.0)]):
        m = synth.surfel_map(n, ref=0, seed=100 + trial).astype(SURFEL_DTYPE)
        m["updateTimes"][rng.random(n) < pdel] = 0         # pre-deleted slots
        m["pz"] += 100.0                                   # far away: the fuse step itself changes nothing else
        m["lastUpdate"] = 0
        g.map_upload(m)
        g.fuse_resident(0, gray, depth, member, pose)
        mg = g.map_download()
        lo, no = o.fuse(0, gray, depth, member, pose, m)
        expect = fuse_map_compact(lo, no)
        assert_surfels_close(mg, expect, f"trial {trial}")
    g.close()


def test_1280x960_sequence(oracle):
    """Config-5 frame size: 19 200 superpixels, three keyframes on the resident map."""
    from manhattanslam_amd import synth, SURFEL_DTYPE
    intr = {k: v * 2 for k, v in synth.TUM1.items()}
    g, o = _mk(intr, 1280, 960)
    m = synth.surfel_map(60000, ref=0).astype(SURFEL_DTYPE)
    g.map_reserve(200000)
    g.map_upload(m)
    o.map_set(m)
    for k in range(3):
        gray, depth, member, pose = synth.surfel_frame(k, 1280, 960, intr=intr, variant="B" if k == 1 else "A")
        g.fuse_resident(k, gray, depth, member, pose)
        o.fuse_map(k, gray, depth, member, pose)
    assert np.array_equal(g.debug_index(), o.index())
    assert_seeds_close(g.debug_seeds(), o.seeds())
    assert_surfels_close(g.map_download(), o.map_get(), "1280x960 map")
    g.close()


def test_1280x960_batched_keyframes(oracle):
    """Five 1280x960 keyframes in ONE batched call on a young map (so that hundreds of seeds spawn per keyframe and stale surfels fall due inside the
    call): k_compact's 19 200-seed flag scan in the default build; under MSL_SF_MERGED=1 (the child-process test below) the compaction wave's
    multi-trip flag loop (5 120 seeds per trip), its scanned listing of many new surfels and its register-kept hand-over list."""
    from manhattanslam_amd import synth, SURFEL_DTYPE
    intr = {k: v * 2 for k, v in synth.TUM1.items()}
    g, o = _mk(intr, 1280, 960)
    g.set_batch_capacity(5)
    m = synth.surfel_map(40000, ref=0, seed=5, min_update_times=1).astype(SURFEL_DTYPE)
    m["lastUpdate"][::7] = -8          # stale with few updates: deleted by the first keyframes of the call
    m["updateTimes"][::7] = 2
    g.map_reserve(300000)
    g.map_upload(m)
    o.map_set(m)
    frames = [synth.surfel_frame(k, 1280, 960, intr=intr, variant="B" if k == 2 else "A") for k in range(5)]
    for k, (gray, depth, member, pose) in enumerate(frames):
        o.fuse_map(k, gray, depth, member, pose)
    g.fuse_resident_batch(list(range(5)), np.stack([f[0] for f in frames]), np.stack([f[1] for f in frames]), np.stack([f[2] for f in frames]),
                          [f[3] for f in frames])
    assert_surfels_close(g.map_download(), o.map_get(), "1280x960 map after a batched call")
    assert np.array_equal(g.debug_index(), o.index())
    t = g.debug_ctr()                  # running totals: [8] new, [9] deleted
    assert t[8] > 500 and t[9] > 1000, t
    g.close()


def test_div100_exact():
    """The division-free x/100.0 used by the cost kernel is the correctly rounded quotient (bit-exact vs IEEE divide)."""
    from manhattanslam_amd import lib
    from manhattanslam_amd._lib import check, ptr
    rng = np.random.default_rng(11)
    x = np.concatenate([rng.uniform(0, 255, 3_000_000), rng.uniform(0, 1, 500_000), np.arange(0, 256, 0.125),
                        np.float32(2.0) ** rng.integers(-60, 60, 100_000) * rng.uniform(1, 2, 100_000)]).astype(np.float32)
    out = np.zeros(len(x), np.float64)
    check(lib.msl_debug_div100(ptr(x), ptr(out), len(x)))
    ref = (x * x).astype(np.float64) / 100.0
    assert np.array_equal(out, ref), np.flatnonzero(out != ref)[:5]


def test_raw_depth16_batches_match_the_float_path(oracle):
    """msl_sf_fuse_resident_batch_d16 (round 5): raw 16-bit depth converted on the device as float(raw) * factor -- Frame::Frame's
    imDepth.convertTo(imDepthScaled, CV_32F, depthMapFactor) (src/Frame.cc:96-97; OpenCV evaluates it in float) -- gives bit for bit the maps, seeds
    and index maps of the float entry point fed with the host-converted images, from host memory and from device memory, and both are the oracle's."""
    import torch
    from manhattanslam_amd import synth, SURFEL_DTYPE
    a = np.float32(1.0) / np.float32(5000.0)                     # mDepthMapFactor = 1.0f / DepthMapFactor (src/Tracking.cc:133-137)
    frames = [synth.clutter_frame(4 * j) for j in range(4)]
    raw = np.stack([synth.depth_u16(f[1]) for f in frames])      # what the sensor delivers
    d32 = raw.astype(np.float32) * a                             # what the reference's host code makes of it
    grays = np.stack([f[0] for f in frames]); member = np.stack([f[2] for f in frames]); poses = [f[3] for f in frames]
    m = synth.surfel_map_dense(120000, ref=0, scene=synth.clutter_scene(), k_lo=-40, k_hi=60, min_update_times=1).astype(SURFEL_DTYPE)
    maps, seeds, idx = [], [], []
    from manhattanslam_amd import lib
    from manhattanslam_amd._lib import check, ptr
    from manhattanslam_amd.surfel import _pose16
    W, H = 640, 480
    rawp = np.full((4, H + 3, W + 24), 0xFFFF, np.uint16)          # rows 2 * (W + 24) bytes apart, frames (H + 3) rows apart, garbage in the padding
    rawp[:, :H, :W] = raw
    for mode in ("float", "raw-host", "raw-host-padded", "raw-device"):
        g, o = _mk(synth.TUM1)
        g.set_batch_capacity(2); g.map_reserve(300000); g.map_upload(m)
        for b in range(2):
            sl = slice(2 * b, 2 * b + 2)
            if mode == "float":
                g.fuse_resident_batch([2 * b, 2 * b + 1], grays[sl], d32[sl], member[sl], poses[sl])
            elif mode == "raw-host":
                g.fuse_resident_batch([2 * b, 2 * b + 1], grays[sl], raw[sl], member[sl], poses[sl], depth_factor=float(a))
            elif mode == "raw-host-padded":   # straight through the C ABI with row / frame strides that are not tight
                refs = np.array([2 * b, 2 * b + 1], np.int32)
                pz = np.ascontiguousarray(np.stack([_pose16(p) for p in poses[sl]]), np.float32)
                gb, mb, rp = np.ascontiguousarray(grays[sl]), np.ascontiguousarray(member[sl]), np.ascontiguousarray(rawp[sl])
                check(lib.msl_sf_fuse_resident_batch_d16(g._h, 2, ptr(refs), ptr(gb), W, W * H, ptr(rp), 2 * (W + 24), 2 * (W + 24) * (H + 3), float(a),
                                                         ptr(mb), 4 * (W // 2), 4 * (W // 2) * (H // 2), 0, ptr(pz)), "d16 padded")
                g.sync()
            else:
                tg, tr, tm = torch.from_numpy(grays[sl]).cuda(), torch.from_numpy(raw[sl].view(np.int16)).cuda(), torch.from_numpy(member[sl]).cuda()
                g.fuse_resident_batch([2 * b, 2 * b + 1], tg, tr, tm, poses[sl], device=True, depth_factor=float(a))
                g.sync()
        maps.append(g.map_download()); seeds.append(g.debug_seeds()); idx.append(g.debug_index())
        if mode == "float":
            o.map_set(m)
            for j in range(4):
                o.fuse_map(j, grays[j], d32[j], member[j], poses[j])
            assert_surfels_close(maps[0], o.map_get(), "float path against the oracle")
        g.close()
    for k in (1, 2, 3):
        assert maps[k].tobytes() == maps[0].tobytes(), k
        assert seeds[k].tobytes() == seeds[0].tobytes(), k
        assert np.array_equal(idx[k], idx[0]), k


def test_rotating_chain_is_the_left_to_right_float_sum():
    """kb_update_seeds evaluates the reference's sequential float sums (depth mean and Huber / Newton numerator, src/SurfelFusion.cpp:486-503) as a
    chain that rotates over 16 lanes (round 5).  On adversarial lists -- every length 0..256, values spread over 20 binades with heavy cancellation,
    so that any other summation order changes the bits -- the chain returns the left-to-right float32 sum bit for bit; with +-inf markers it adds the
    DOUBLE constant +-0.4 the way the reference's promotion does."""
    from manhattanslam_amd import lib
    from manhattanslam_amd._lib import check, ptr
    rng = np.random.default_rng(5)
    lens = np.concatenate([np.arange(257), rng.integers(0, 257, 767)]).astype(np.int32)
    L = len(lens)
    x = (rng.uniform(-1, 1, (L, 256)) * np.float32(2.0) ** rng.integers(-10, 10, (L, 256))).astype(np.float32)
    x[5] = 0.0; x[6, ::2] = -0.0                                   # zeros of both signs
    x[7, :200] = np.tile(np.array([1e8, 1.0, -1e8, 1.0], np.float32), 50)   # order-sensitive cancellation
    out = np.zeros(L, np.float32)
    check(lib.msl_debug_chain_sum(ptr(x), ptr(lens), L, 0, ptr(out)))
    ref = np.zeros(L, np.float32)
    for q in range(L):
        s = np.float32(0.0)
        for e in range(lens[q]):
            s = np.float32(s + x[q, e])
        ref[q] = s
    assert np.array_equal(out.view(np.int32), ref.view(np.int32)), np.flatnonzero(out.view(np.int32) != ref.view(np.int32))[:5]
    # order sensitivity of the inputs: a pairwise (tree) sum differs on most lists, so the equality above does discriminate
    tree = np.array([np.sum(x[q, :lens[q]], dtype=np.float32) for q in range(L)])
    assert (tree.view(np.int32) != ref.view(np.int32)).mean() > 0.3
    # Huber tails: +-inf marks an element whose contribution is (float)((double)s +- 0.4)
    xh = x.copy()
    mark = rng.random((L, 256)) < 0.15
    xh[mark] = np.where(rng.random(int(mark.sum())) < 0.5, np.float32(np.inf), np.float32(-np.inf))
    check(lib.msl_debug_chain_sum(ptr(xh), ptr(lens), L, 1, ptr(out)))
    for q in range(L):
        s = np.float32(0.0)
        for e in range(lens[q]):
            t = xh[q, e]
            s = np.float32(np.float64(s) + (0.4 if t > 0 else -0.4)) if np.isinf(t) else np.float32(s + t)
        ref[q] = s
    assert np.array_equal(out.view(np.int32), ref.view(np.int32)), np.flatnonzero(out.view(np.int32) != ref.view(np.int32))[:5]


def test_map_maintenance_matches_literal_loops(oracle):
    """moveAddSurfels detach / re-attach and the Stop() export filter on the resident map."""
    import ctypes as C
    from manhattanslam_amd import synth, SURFEL_DTYPE
    from tests.oracle_lib import load, _p
    d = load().dll
    d.mslo_map_detach.restype = C.c_size_t; d.mslo_map_detach.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    d.mslo_map_export.restype = C.c_size_t; d.mslo_map_export.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    g, o = _mk(synth.TUM1)
    m = synth.surfel_map(70000, ref=8).astype(SURFEL_DTYPE)          # lastUpdate in 0..8, updateTimes 1..20
    m["updateTimes"][::17] = 0                                       # some already deleted slots
    g.map_upload(m)
    ref_map = m.copy()
    for pose in (3, 8, 100):
        buf = np.zeros(len(ref_map), SURFEL_DTYPE)
        k = d.mslo_map_detach(_p(ref_map), len(ref_map), pose, _p(buf))
        got = g.map_detach(pose)
        assert len(got) == k and got.tobytes() == buf[:k].tobytes(), pose
        assert g.map_download().tobytes() == ref_map.tobytes()
    buf = np.zeros(len(ref_map), SURFEL_DTYPE)
    k = d.mslo_map_export(_p(ref_map), len(ref_map), 5, _p(buf))
    exp = g.map_export(5)
    assert len(exp) == k and exp.tobytes() == buf[:k].tobytes() and 0 < k < len(ref_map)
    extra = synth.surfel_map(1234, ref=9, seed=5).astype(SURFEL_DTYPE)
    g.map_append(extra)                                              # mvLocalSurfels.insert(end, attachedSurfels...)
    assert g.map_download().tobytes() == np.concatenate([ref_map, extra]).tobytes()
    # and fusion keeps working on the maintained map
    gray, depth, member, pose = synth.surfel_frame(9)
    o.map_set(np.concatenate([ref_map, extra]))
    g.fuse_resident(9, gray, depth, member, pose)
    o.fuse_map(9, gray, depth, member, pose)
    assert_surfels_close(g.map_download(), o.map_get(), "after maintenance")
    g.close()


def test_peac_block_stats_match_oracle(oracle):
    """organised cloud + initial PEAC block statistics, bit-identical FP64 (batched, strict and loose init)."""
    from manhattanslam_amd import synth, peac
    from tests.oracle_lib import peac_block_stats
    I = synth.TUM1
    rng = np.random.default_rng(5)
    frames = []
    for k in range(3):
        _, depth, _, _ = synth.surfel_frame(k, variant="B" if k == 1 else "A")
        d16 = np.clip(np.round(depth * 5000.0), 0, 65535).astype(np.uint16)
        d16[rng.random(d16.shape) < 0.002] = 0                      # missing returns
        if k == 2:
            d16[100:300, 200:420] += 5000                           # a box 1 m further away: depth discontinuities
        frames.append(d16)
    frames = np.stack(frames)
    factor = np.float32(1.0 / 5000.0)
    for loose in (False, True):
        cloud, st = peac.block_stats(frames, I["fx"], I["fy"], I["cx"], I["cy"], factor, init_loose=loose)
        assert st.shape == (3, 24 * 32)
        for k in range(3):
            c_ref, s_ref = peac_block_stats(frames[k], I["fx"], I["fy"], I["cx"], I["cy"], factor, init_loose=loose)
            assert cloud[k].tobytes() == c_ref.tobytes()
            assert st[k].tobytes() == s_ref.tobytes(), (loose, k)
        assert (st["nouse"] == 0).sum() > 0 and (loose or (st["nouse"] == 1).sum() > 0)
    assert (st[2]["nouse"] == 1).sum() > 0                          # loose init still rejects the depth steps of frame 2
    # odd sizes and a non-default window
    d = frames[0][:431, :517]
    cloud, st = peac.block_stats(d, I["fx"], I["fy"], I["cx"], I["cy"], factor, window=(8, 12))
    c_ref, s_ref = peac_block_stats(d, I["fx"], I["fy"], I["cx"], I["cy"], factor, window=(8, 12))
    assert cloud[0].tobytes() == c_ref.tobytes() and st[0].tobytes() == s_ref.tobytes()


def test_full_size_map_batched_matches_oracle(oracle):
    """BASELINE config 3 at full size: ~1 M live surfels, a batch of keyframes on the resident map vs the oracle run keyframe by keyframe."""
    from manhattanslam_amd import synth, SURFEL_DTYPE
    g, o = _mk(synth.TUM1)
    n, F = 1_000_000, 4
    m = synth.surfel_map(n, ref=0, min_update_times=1).astype(SURFEL_DTYPE)     # includes young surfels: real deletions
    g.set_batch_capacity(F)
    g.map_reserve(n + 100_000)
    g.map_upload(m)
    o.map_set(m)
    frames = [synth.surfel_frame(k, variant="B" if k == 2 else "A") for k in range(F)]
    refs = np.arange(7, 7 + F)                                                  # ref - lastUpdate > 5 for part of the map
    g.fuse_resident_batch(refs, np.stack([f[0] for f in frames]), np.stack([f[1] for f in frames]), np.stack([f[2] for f in frames]),
                          [f[3] for f in frames])
    for k in range(F):
        o.fuse_map(int(refs[k]), *frames[k])
    mg, mo = g.map_download(), o.map_get()
    assert len(mg) == len(mo)
    assert_surfels_close(mg, mo, "1M map after 4 keyframes")
    c = g.counters()
    assert c["n_live_after"] == len(mo) and c["n_deleted"] > 0
    g.close()


@pytest.mark.parametrize("w,h", [(328, 248), (320, 240), (72, 64)])
def test_odd_seed_lattices_and_small_images(oracle, w, h):
    """Seed lattices with odd extents (41 x 31: partial 2x2 seed blocks, partial tiles) and tiny images, two keyframes."""
    from manhattanslam_amd import synth, SURFEL_DTYPE
    intr = {k: v * (w / 640.0) for k, v in synth.TUM1.items()}
    g, o = _mk(intr, w, h)
    m = synth.surfel_map(20000, ref=0).astype(SURFEL_DTYPE)
    g.map_upload(m)
    o.map_set(m)
    for k in range(2):
        gray, depth, member, pose = synth.surfel_frame(k, w, h, intr=intr, variant="B" if k == 1 else "A")
        g.fuse_resident(k, gray, depth, member, pose)
        o.fuse_map(k, gray, depth, member, pose)
        assert np.array_equal(g.debug_index(), o.index()), (w, h, k)
        assert_seeds_close(g.debug_seeds(), o.seeds())
    assert_surfels_close(g.map_download(), o.map_get(), f"{w}x{h} map")
    g.close()


def test_large_image_uses_relaxation_fallback(oracle):
    """2048 x 1536: 49 152 seeds exceed the LDS-resident relaxation kernel, so the multi-launch fallback runs (two keyframes)."""
    from manhattanslam_amd import synth, SURFEL_DTYPE
    w, h = 2048, 1536
    intr = {k: v * (w / 640.0) for k, v in synth.TUM1.items()}
    g, o = _mk(intr, w, h)
    m = synth.surfel_map(30000, ref=0).astype(SURFEL_DTYPE)
    g.map_reserve(200000)
    g.map_upload(m)
    o.map_set(m)
    for k in range(2):
        gray, depth, member, pose = synth.surfel_frame(k, w, h, intr=intr, variant="B" if k == 1 else "A")
        g.fuse_resident(k, gray, depth, member, pose)
        o.fuse_map(k, gray, depth, member, pose)
    assert np.array_equal(g.debug_index(), o.index())
    assert_seeds_close(g.debug_seeds(), o.seeds())
    assert_surfels_close(g.map_download(), o.map_get(), "2048x1536 map")
    g.close()


def test_handles_of_different_sizes_coexist(oracle):
    """A large-image handle keeps working after a small-image handle is created (per-function launch attributes are shared)."""
    from manhattanslam_amd import synth, SURFEL_DTYPE
    big_i = {k: v * 2 for k, v in synth.TUM1.items()}
    gb, ob = _mk(big_i, 1280, 960)
    gs, os_ = _mk(synth.TUM1)                                    # created second: must not shrink anything the first one needs
    m = synth.surfel_map(20000, ref=0).astype(SURFEL_DTYPE)
    for g, o, (w, h, intr) in ((gb, ob, (1280, 960, big_i)), (gs, os_, (640, 480, synth.TUM1))):
        g.map_upload(m)
        o.map_set(m)
        for k in range(2):
            f = synth.surfel_frame(k, w, h, intr=intr)
            g.fuse_resident(k, *f)
            o.fuse_map(k, *f)
        assert_surfels_close(g.map_download(), o.map_get(), f"{w}x{h}")
    gb.close(); gs.close()


def test_config4_icl_live_map_negative_fy(oracle):
    """BASELINE config 4 as a parity case: ICL intrinsics (fy = -480, Example/ICL.yaml:8-11) on a NON-empty resident map with plane
    membership variant B, four keyframes: k_fuse's projection / cameraF = (|fx| + |fy|) / 2 path (src/SurfelFusion.cpp:75-78,
    204-221) under negative fy, incl. deletions, updates, new surfels and compaction."""
    from manhattanslam_amd import synth, SURFEL_DTYPE
    g, o = _mk(synth.ICL)
    m = synth.surfel_map(150000, ref=0).astype(SURFEL_DTYPE)
    g.map_reserve(300000)
    g.map_upload(m)
    o.map_set(m)
    tot_upd = tot_del = 0
    for k in range(4):
        gray, depth, member, pose = synth.surfel_frame(k, intr=synth.ICL, variant="B")
        g.fuse_resident(k, gray, depth, member, pose)
        n_new = o.fuse_map(k, gray, depth, member, pose)
        c = g.counters()
        assert c["n_new"] == n_new
        tot_upd += c["n_updated"]; tot_del += c["n_deleted"]
        assert np.array_equal(g.debug_index(), o.index())
        assert_seeds_close(g.debug_seeds(), o.seeds())
        assert_surfels_close(g.map_download(), o.map_get(), f"ICL map after keyframe {k}")
    assert tot_upd > 10000 and tot_del > 100      # the live map really is fused under fy < 0
    g.close()


def test_keyframe_every_k_frame_step(oracle):
    """fuse_resident_batch(frame_step=k): keyframe j = frame j*k of a device-resident sequence (bench config 4's cadence)."""
    import torch
    from manhattanslam_amd import synth, SURFEL_DTYPE
    g, o = _mk(synth.ICL)
    g.set_batch_capacity(3)
    m = synth.surfel_map(60000, ref=0).astype(SURFEL_DTYPE)
    g.map_reserve(200000); g.map_upload(m); o.map_set(m)
    frames = [synth.surfel_frame(k, intr=synth.ICL, variant="B") for k in range(5)]
    dg, dd = (torch.from_numpy(np.stack([f[i] for f in frames])).cuda() for i in (0, 1))
    dm = torch.from_numpy(frames[0][2]).cuda()
    g.fuse_resident_batch([0, 1, 2], dg, dd, dm, [frames[j][3] for j in (0, 2, 4)], device=True, member_shared=True, frame_step=2)
    for r, j in enumerate((0, 2, 4)):
        o.fuse_map(r, frames[j][0], frames[j][1], frames[j][2], frames[j][3])
    assert_surfels_close(g.map_download(), o.map_get(), "map after keyframes 0, 2, 4")
    g.close()


def test_resident_map_grows_without_reserve(oracle):
    """The reference's mvLocalSurfels is an unbounded std::vector: fusing keyframes past the initial 65 536-slot capacity without
    any msl_sf_map_reserve call must keep every new surfel (the map silently stopped growing)."""
    from manhattanslam_amd import synth, SURFEL_DTYPE
    g, o = _mk(synth.TUM1)
    # 60 000 stable surfels far outside the fusion range (never updated, never deleted): the map can only grow
    m = synth.surfel_map(60000, ref=0, min_update_times=5).astype(SURFEL_DTYPE)
    m["px"] += 100.0; m["py"] += 100.0; m["pz"] += 100.0
    g.map_upload(m)          # capacity = what create/upload chose (65 536); no reserve
    o.map_set(m)
    for k in range(4):       # views 60 degrees apart: each keyframe spawns a few thousand new surfels
        gray, depth, member, pose = synth.surfel_frame(120 * k)
        g.fuse_resident(k, gray, depth, member, pose)
        o.fuse_map(k, gray, depth, member, pose)
    mo = o.map_get()
    assert len(mo) > 65536 + 2000, len(mo)
    assert_surfels_close(g.map_download(), mo, "map grown past its initial capacity")
    # batched form: several keyframes enqueued at once must reserve for all of them up front
    g.set_batch_capacity(4)
    frames = [synth.surfel_frame(60 + 120 * k) for k in range(8)]
    for b in range(2):
        fr = frames[4 * b:4 * b + 4]
        g.fuse_resident_batch([4 + 4 * b + j for j in range(4)], np.stack([f[0] for f in fr]), np.stack([f[1] for f in fr]),
                              np.stack([f[2] for f in fr]), [f[3] for f in fr])
        for j, f in enumerate(fr):
            o.fuse_map(4 + 4 * b + j, f[0], f[1], f[2], f[3])
    mo = o.map_get()
    assert len(mo) > 80000, len(mo)
    assert_surfels_close(g.map_download(), mo, "map grown by batches")
    g.close()


def test_export_ply_matches_map_export(oracle, tmp_path):
    """System::saveSurfels layout (src/System.cc:296-382): vertex lines = the updateTimes >= 5 surfels in map order + the inactive ones."""
    from manhattanslam_amd import synth, SURFEL_DTYPE
    g, _ = _mk(synth.TUM1)
    m = synth.surfel_map(5000, ref=0).astype(SURFEL_DTYPE)
    g.map_upload(m)
    ina = synth.surfel_map(40, ref=0, seed=99).astype(SURFEL_DTYPE)
    ina["px"][3] = np.nan                                    # skipped (:311-312)
    path = tmp_path / "Surfels.ply"
    g.export_ply(path, 5, ina)
    lines = open(path).read().splitlines()
    hdr = lines.index("end_header")
    want = np.concatenate([g.map_export(5), ina[~np.isnan(ina["px"])]])
    assert lines[0] == "ply" and lines[1] == "format ascii 1.0" and lines[2] == f"element vertex {len(want)}"
    assert lines[3:15] == [f"property float {p}" for p in ("x", "y", "z", "nx", "ny", "nz")] + [f"property uchar {p}" for p in ("red", "green", "blue", "alpha")] + [
        "property float quality", "property float radius"]
    body = np.array([[float(v) for v in ln.split()] for ln in lines[hdr + 1:hdr + 1 + len(want)]])
    assert body.shape == (len(want), 12)
    for col, f in enumerate(("px", "py", "pz", "nx", "ny", "nz")):
        assert np.allclose(body[:, col], want[f], rtol=2e-5, atol=1e-6)          # default ostream precision: 6 significant digits
    assert np.array_equal(body[:, 6].astype(int), want["r"] & 255) and np.all(body[:, 9] == 1)
    assert np.allclose(body[:, 10], want["weight"], rtol=2e-5) and np.allclose(body[:, 11], want["size"] * 1000, rtol=2e-5)
    cam = lines[hdr + 1 + len(want)].split()
    assert len(cam) == 21 and int(cam[17]) == len(want) and len(lines) == hdr + 2 + len(want)
    g.close()


def test_snapshot_restore_replays_identically(oracle):
    """msl_sf_map_snapshot / msl_sf_map_restore (bench.py's stationary passes): after a restore the same keyframes give the same map,
    and that map is the oracle's."""
    from manhattanslam_amd import synth, SURFEL_DTYPE
    g, o = _mk(synth.TUM1)
    m = synth.surfel_map(30000, ref=0, min_update_times=5).astype(SURFEL_DTYPE)
    g.map_upload(m)
    g.map_snapshot()
    o.map_set(m)
    frames = [synth.surfel_frame(k) for k in range(3)]
    g.set_batch_capacity(3)
    args = (np.stack([f[0] for f in frames]), np.stack([f[1] for f in frames]), np.stack([f[2] for f in frames]), [f[3] for f in frames])
    g.fuse_resident_batch([0, 1, 2], *args)
    first = g.map_download()
    for k, f in enumerate(frames):
        o.fuse_map(k, f[0], f[1], f[2], f[3])
    assert_surfels_close(first, o.map_get(), "first pass")
    assert len(first) != len(m)
    g.map_restore()                       # asynchronous: ordered behind the keyframes above on the map stream
    assert g.map_size() == len(m)
    assert g.map_download().tobytes() == m.tobytes()
    g.map_restore()
    g.fuse_resident_batch([0, 1, 2], *args)
    assert g.map_download().tobytes() == first.tobytes()
    # a snapshot larger than the current allocation (after an upload of a small map) is restored into a regrown map
    g.map_upload(m[:10])
    g.map_restore()
    assert g.map_download().tobytes() == m.tobytes()
    g.close()


def test_upload_after_resident_batches_keeps_growing(oracle):
    """a live-count snapshot recorded by an earlier resident batch must not lower the host-side bound below the size of a
    map uploaded afterwards; otherwise the grow-before-overflow check is skipped and k_compact drops new surfels (deferred error 20)."""
    from manhattanslam_amd import synth, SURFEL_DTYPE
    g, o = _mk(synth.TUM1)
    small = synth.surfel_map(2000, ref=0, min_update_times=5).astype(SURFEL_DTYPE)
    g.map_upload(small)
    gray, depth, member, pose = synth.surfel_frame(0)
    g.set_batch_capacity(2)
    g.fuse_resident_batch([0, 1], np.stack([gray, gray]), np.stack([depth, depth]), np.stack([member, member]), [pose, pose])   # leaves an async count snapshot behind (no sync here: msl_sf_sync would clear it)
    # a larger map whose size is just below the capacity upload() chooses; all of it far outside the fusion range, so it only grows
    big = synth.surfel_map(60000, ref=0, min_update_times=5).astype(SURFEL_DTYPE)
    big["px"] += 100.0; big["py"] += 100.0; big["pz"] += 100.0
    g.map_upload(big)
    o.map_set(big)
    for k in range(4):
        fr = [synth.surfel_frame(120 * k + 60 * j) for j in range(2)]
        g.fuse_resident_batch([2 * k, 2 * k + 1], np.stack([f[0] for f in fr]), np.stack([f[1] for f in fr]), np.stack([f[2] for f in fr]), [f[3] for f in fr])
        for j, f in enumerate(fr):
            o.fuse_map(2 * k + j, f[0], f[1], f[2], f[3])
    mo = o.map_get()
    assert len(mo) > 65536 + 2000, len(mo)      # past the 65 536-slot capacity the upload kept
    assert_surfels_close(g.map_download(), mo, "map uploaded after resident batches, grown past the old capacity")
    g.close()


def test_wide_rgb_values_survive_the_byte_packed_cold_record(oracle):
    """The resident map stores r, g, b as three bytes (they come from a cv::Vec3b, src/SurfelFusion.cpp:484, 551); an uploaded surfel
    whose ints do not fit a byte keeps them exactly (side array), through upload / fuse / compaction moves / detach / append / snapshot."""
    from manhattanslam_amd import synth, SURFEL_DTYPE
    g, o = _mk(synth.TUM1)
    m = synth.surfel_map(30000, ref=2, min_update_times=1).astype(SURFEL_DTYPE)
    rng = np.random.default_rng(5)
    wide = rng.choice(len(m), 3000, replace=False)
    m["r"][wide] = rng.integers(-2**31, 2**31 - 1, len(wide)); m["g"][wide[::2]] = 256; m["b"][wide[::3]] = -1
    g.map_upload(m)
    assert g.map_download().tobytes() == m.tobytes()
    g.map_snapshot()
    o.map_set(m)
    for k in (2, 9):      # keyframe 9 deletes stale surfels: tail moves carry wide records along
        gray, depth, member, pose = synth.surfel_frame(k)
        g.fuse_resident(k, gray, depth, member, pose)
        o.fuse_map(k, gray, depth, member, pose)
    mo = o.map_get()
    mg = g.map_download()
    assert_surfels_close(mg, mo, "map with wide r, g, b")
    assert ((mo["r"] < 0) | (mo["r"] > 255)).sum() > 100
    det = g.map_detach(9)
    assert len(det) and np.array_equal(det["r"], mo["r"][(mo["lastUpdate"] == 9) & (mo["updateTimes"] > 0)])
    g.map_append(m[wide[:50]])
    assert np.array_equal(g.map_download()["r"][-50:], m["r"][wide[:50]])
    g.map_restore()
    assert g.map_download().tobytes() == m.tobytes()
    g.close()


def test_host_image_batches_with_shared_membership_and_strided_frames(oracle):
    """The streaming shapes of msl_sf_fuse_resident_batch with MSL_MEM_HOST: tightly packed frame arrays (one copy per image kind on the
    copy stream), one membership image shared by the call (staged once), and a frame step > 1 (per-frame copies) -- all equal the oracle."""
    from manhattanslam_amd import synth, SURFEL_DTYPE
    g, o = _mk(synth.TUM1)
    m = synth.surfel_map(20000, ref=0, min_update_times=5).astype(SURFEL_DTYPE)
    g.map_upload(m); o.map_set(m)
    frames = [synth.surfel_frame(k, variant="B") for k in range(8)]
    grays = np.stack([f[0] for f in frames]); depths = np.stack([f[1] for f in frames]); member = frames[0][2]
    g.set_batch_capacity(4)
    g.fuse_resident_batch([0, 1, 2, 3], grays, depths, member, [frames[j][3] for j in range(4)], member_shared=True)                 # packed
    g.fuse_resident_batch([4, 5, 6, 7], grays[4:], depths[4:], member, [frames[j][3] for j in range(4, 8)], member_shared=True)       # other slot set
    g.fuse_resident_batch([8, 9, 10, 11], grays, depths, member, [frames[j][3] for j in (0, 2, 4, 6)], member_shared=True, frame_step=2)   # strided
    for r, j in enumerate(list(range(8)) + [0, 2, 4, 6]):
        o.fuse_map(r, frames[j][0], frames[j][1], member, frames[j][3])
    assert_surfels_close(g.map_download(), o.map_get(), "host-image batches")
    g.close()


@pytest.mark.parametrize("raw16", [False, True])
def test_six_host_image_calls_in_flight(oracle, raw16):
    """Streaming from host memory (bench.py --io host): six calls enqueued back to back without a sync, every call with its own frames.  The H2D copies of
    call k run on the copy stream and (round 6) wait only for the SUPERPIXEL stage of call k - 2 -- the last reader of the set's staged images --, not
    for its map stage, so the link runs ahead of the map chain; a copy that overtook a reader, or a superpixel stage that overtook the map stage of the
    slot arrays it rewrites, would show as a map that differs from the oracle's."""
    from manhattanslam_amd import synth, SURFEL_DTYPE
    g, o = _mk(synth.TUM1)
    m = synth.surfel_map_dense(150_000, ref=0).astype(SURFEL_DTYPE)
    g.set_batch_capacity(4)
    g.map_reserve(len(m) + 100_000)
    g.map_upload(m); o.map_set(m)
    nk = 24
    frames = [synth.surfel_frame(k, variant="B" if k % 5 == 3 else "A") for k in range(nk)]
    factor = float(np.float32(1.0) / np.float32(5000.0))
    d16 = [np.clip(np.rint(f[1] * 5000.0), 0, 65535).astype(np.uint16) for f in frames]
    depth_f = [(d.astype(np.float32) * np.float32(factor)) if raw16 else f[1] for d, f in zip(d16, frames)]      # what the device conversion yields
    member = frames[0][2]
    for c in range(nk // 4):
        sl = slice(4 * c, 4 * c + 4)
        g.fuse_resident_batch(np.arange(4 * c, 4 * c + 4), np.stack([f[0] for f in frames[sl]]), np.stack(d16[sl]) if raw16 else np.stack([f[1] for f in frames[sl]]),
                              member, [f[3] for f in frames[sl]], member_shared=True, depth_factor=factor if raw16 else None)
    for k in range(nk):
        o.fuse_map(k, frames[k][0], depth_f[k], member, frames[k][3])
    assert_surfels_close(g.map_download(), o.map_get(), "six host-image calls in flight")
    g.close()


def test_map_outgrows_the_fuse_grid_inside_one_call(oracle):
    """k_fuse's grid covers the host's last KNOWN live count plus 2 keyframes' worth of seeds; a call whose keyframes add more than that (an
    empty map and 16 fresh views) is still covered because the kernel is grid-stride over the device's live count."""
    from manhattanslam_amd import synth, SURFEL_DTYPE
    g, o = _mk(synth.TUM1)
    g.map_upload(np.zeros(0, SURFEL_DTYPE)); o.map_set(np.zeros(0, SURFEL_DTYPE))
    frames = [synth.surfel_frame(45 * k) for k in range(16)]          # 45 degrees apart: every keyframe spawns thousands of new surfels
    g.set_batch_capacity(16)
    frames = [synth.surfel_frame(23 * k) for k in range(16)]
    refs = [7] * 16      # one reference index for all of them (it is an opaque int): nothing goes stale, the map only grows
    g.fuse_resident_batch(refs, np.stack([f[0] for f in frames]), np.stack([f[1] for f in frames]), np.stack([f[2] for f in frames]),
                          [f[3] for f in frames])
    for k, f in enumerate(frames):
        o.fuse_map(refs[k], f[0], f[1], f[2], f[3])
    mo = o.map_get()
    assert len(mo) > 2 * g.nseeds + 8192, len(mo)      # well beyond the grid of the call's later launches
    assert_surfels_close(g.map_download(), mo, "map grown past the fuse grid inside one call")
    g.close()


@pytest.mark.parametrize("w,h", [(324, 244), (322, 246), (333, 251), (330, 243), (641, 479)])
def test_sizes_that_are_not_multiples_of_eight(oracle, w, h):
    """The reference truncates (spWidth = width / SP_SIZE, src/SurfelFusion.cpp:29-38): the strips right of / below the last whole cell belong
    to no superpixel cell but take part in updatePixels, the seed windows (flat-index wrap-around included) and the fusion.  Odd sizes, widths
    with W mod 4 in {1, 2, 3} (window quads that straddle the right edge) and both execution modes against the oracle."""
    from manhattanslam_amd import synth, SURFEL_DTYPE
    intr = synth.scaled_intrinsics(synth.TUM1, w)
    g, o = _mk(intr, w, h)
    assert g.nseeds == (w // 8) * (h // 8)
    local = synth.surfel_map(20000, ref=3).astype(SURFEL_DTYPE)
    frames = [synth.surfel_frame(3 + k, w=w, h=h, intr=intr, variant="B") for k in range(3)]
    gray, depth, member, pose = frames[0]
    assert member.shape == ((h + 1) // 2, (w + 1) // 2)
    lo, no = o.fuse(3, gray, depth, member, pose, local)
    lg = local.copy()
    ng = g.fuseInitializeMap(3, gray, depth, member, pose, lg)
    assert np.array_equal(g.debug_index(), o.index())
    assert_seeds_close(g.debug_seeds(), o.seeds())
    assert_surfels_close(lg, lo, "local")
    assert_surfels_close(ng, no, "new")
    assert len(ng) > 20 and (lg["lastUpdate"] == 3).sum() > 100
    # resident, batched
    g.map_upload(local); o.map_set(local)
    g.set_batch_capacity(3)
    g.fuse_resident_batch([3, 4, 5], np.stack([f[0] for f in frames]), np.stack([f[1] for f in frames]), np.stack([f[2] for f in frames]), [f[3] for f in frames])
    for k, f in enumerate(frames):
        o.fuse_map(3 + k, f[0], f[1], f[2], f[3])
    assert_surfels_close(g.map_download(), o.map_get(), "resident map")
    g.close()


def test_host_vector_unchanged_hint_and_touched_range_download(oracle):
    """msl_sf_fuse_ex: with MSL_SF_LOCAL_UNCHANGED the caller's vector is not uploaded again (the device copy of the previous call is fused into),
    and only the stretches of the vector that hold surfels the keyframe touched come back.  Four keyframes in a row on one vector, the hint given
    from the second on, against the oracle; then the caller edits the vector (no hint: full upload), then a hint with a changed length (ignored)."""
    from manhattanslam_amd import synth, SURFEL_DTYPE
    g, o = _mk(synth.TUM1)
    m = synth.surfel_map_dense(300000, ref=0, k_lo=-150, k_hi=214, min_update_times=1, flip=0.02, floating=0.01).astype(SURFEL_DTYPE)
    far = np.arange(len(m)) >= 200000
    m["px"][far] += 100.0; m["pz"][far] += 100.0        # a third of the map is out of range for good: its sub-blocks are never sent back
    m["updateTimes"][far] = 9
    lg, lo = m.copy(), m.copy()
    for k in range(4):
        gray, depth, member, pose = synth.surfel_frame(2 * k, variant="B" if k == 1 else "A")
        lo, no = o.fuse(k, gray, depth, member, pose, lo)
        ng = g.fuseInitializeMap(k, gray, depth, member, pose, lg, local_unchanged=k > 0)
        assert_surfels_close(lg, lo, f"local after keyframe {k}")
        assert_surfels_close(ng, no, f"new after keyframe {k}")
        assert lg[far].tobytes() == m[far].tobytes()
    assert (lg["lastUpdate"] == 3).sum() > 20000
    # the caller edits its vector: no hint -> the edit is seen
    lg["pz"][:1000] += 50.0; lo["pz"][:1000] += 50.0
    gray, depth, member, pose = synth.surfel_frame(9)
    lo, no = o.fuse(4, gray, depth, member, pose, lo)
    ng = g.fuseInitializeMap(4, gray, depth, member, pose, lg)
    assert_surfels_close(lg, lo, "after an edit, no hint")
    # a hint that cannot hold (the vector grew): ignored, full upload
    extra = synth.surfel_map(5000, ref=4, seed=3).astype(SURFEL_DTYPE)
    lg = np.concatenate([lg, extra]); lo = np.concatenate([lo, extra])
    gray, depth, member, pose = synth.surfel_frame(11)
    lo, no = o.fuse(5, gray, depth, member, pose, lo)
    ng = g.fuseInitializeMap(5, gray, depth, member, pose, lg, local_unchanged=True)
    assert_surfels_close(lg, lo, "grown vector with a stale hint")
    assert_surfels_close(ng, no, "new")
    g.close()


def test_host_vector_sparse_change_list(oracle):
    """msl_sf_fuse_ex's other way back: a map in no particular order (every sub-block touched, few surfels each) returns a compact
    {index, record} list instead of whole stretches; also with surfels that carried lastUpdate == ref before the call (listed, unchanged)."""
    from manhattanslam_amd import synth, SURFEL_DTYPE
    g, o = _mk(synth.TUM1)
    m = synth.surfel_map(400000, ref=1, min_update_times=5).astype(SURFEL_DTYPE)      # area-uniform room map: ~6 % in view, lastUpdate in -7..1
    lg, lo = m.copy(), m.copy()
    for k in (1, 2, 3):
        gray, depth, member, pose = synth.surfel_frame(k)
        lo, no = o.fuse(k, gray, depth, member, pose, lo)
        ng = g.fuseInitializeMap(k, gray, depth, member, pose, lg, local_unchanged=k > 1)
        assert_surfels_close(lg, lo, f"local after keyframe {k}")
        assert_surfels_close(ng, no, f"new after keyframe {k}")
    assert 5000 < (lg["lastUpdate"] == 3).sum() < 50000
    g.close()


def test_classic_chain_gives_identical_maps():
    """The parity suite pins the DEFERRED compaction (tests/conftest.py: MSL_SF_DEFER=1; round 5: one k_fuse launch per keyframe, new surfels appended
    physically, the window's placements and tail moves replayed at its end, msl_sf_map.hip).  MSL_SF_DEFER=0 sends every keyframe through the classic pair
    k_fuse + k_compact instead.  The flag is read once per process, so the batched / resident parity tests run again in a child process with
    it set: both chains leave the oracle's maps, counters and new-surfel lists."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env["MSL_SF_DEFER"] = "0"
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-m", "gpu", "-x", "-p", "no:cacheprovider",
                        os.path.join(root, "tests", "test_surfel_gpu.py"), os.path.join(root, "tests", "test_clutter_gpu.py"),
                        "-k", "batched or full_size or grows or resident_sequence or snapshot or outgrows or dense_in_view or keyframe_every or compaction or wide or moving"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout


def test_default_policy_picks_the_chain_and_keeps_parity():
    """tests/conftest.py pins MSL_SF_DEFER=1 and test_classic_chain_gives_identical_maps pins 0; bench.py -- and any caller -- runs with NEITHER: the library
    then chooses per batch (msl_surfel.hip, run_batch).  tests/policy_child.py runs that unset-environment path in a child process: a handle with its own
    two streams (bench.py's shape) takes the classic chain for every keyframe, a handle on one caller stream takes deferred windows, and falls back to
    the classic chain once the churn estimate flips -- all three against the oracle, batch by batch."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k != "MSL_SF_DEFER"}
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "policy_child.py")], cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "policy ok" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def test_wide_update_counters_survive_the_packed_hot_record(oracle):
    """The resident hot record packs updateTimes (11 bits) and lastUpdate (20 bits) into one word (round 5); values outside those ranges --
    negative reference indices, more than 2047 fusions, indices beyond a million -- keep their exact ints in a side array, through upload,
    classic and deferred keyframes (fusion increments, stale deletions, tail moves), detach / append and snapshot / restore."""
    from manhattanslam_amd import synth, SURFEL_DTYPE
    g, o = _mk(synth.TUM1)
    m = synth.surfel_map(30000, ref=2000000, min_update_times=1).astype(SURFEL_DTYPE)   # lastUpdate around two million: every record is wide
    rng = np.random.default_rng(7)
    big = rng.choice(len(m), 4000, replace=False)
    m["updateTimes"][big] = rng.integers(2040, 5000, len(big))     # around and beyond the 11-bit limit
    m["lastUpdate"][big[::5]] = -3                                  # negative reference index (stale: deleted when updateTimes < 5 -- none here)
    m["updateTimes"][big[1::7]] = 2047                              # the largest packed value: one more fusion makes it wide
    g.map_upload(m)
    assert g.map_download().tobytes() == m.tobytes()
    g.map_snapshot()
    o.map_set(m)
    g.set_batch_capacity(4)
    refs = [2000002, 2000003, 2000009, 2000010]                     # 2000009 deletes what went stale
    frames = [synth.surfel_frame(k) for k in (2, 3, 9, 10)]
    g.fuse_resident_batch(refs, np.stack([f[0] for f in frames]), np.stack([f[1] for f in frames]), np.stack([f[2] for f in frames]), [f[3] for f in frames])
    for r, f in zip(refs, frames):
        o.fuse_map(r, f[0], f[1], f[2], f[3])
    mo = o.map_get()
    assert_surfels_close(g.map_download(), mo, "map with wide updateTimes / lastUpdate")
    assert (mo["updateTimes"] > 2047).sum() > 100 and (mo["lastUpdate"] > (1 << 20)).sum() > 1000
    # a narrow map on the same handle: small reference indices, the packed form
    m2 = synth.surfel_map(20000, ref=2, min_update_times=1).astype(SURFEL_DTYPE)
    g.map_upload(m2); o.map_set(m2)
    frames = [synth.surfel_frame(k) for k in (2, 3, 9)]
    g.fuse_resident_batch([2, 3, 9], np.stack([f[0] for f in frames]), np.stack([f[1] for f in frames]), np.stack([f[2] for f in frames]), [f[3] for f in frames])
    for r, f in zip([2, 3, 9], frames):
        o.fuse_map(r, f[0], f[1], f[2], f[3])
    assert_surfels_close(g.map_download(), o.map_get(), "narrow map after a wide one")
    g.map_restore()
    assert g.map_download().tobytes() == m.tobytes()
    g.close()


@pytest.mark.parametrize("n,shares,tail_heavy,same_view", [
    (20000, (0.3, 0.2, 0.1), False, False), (5000, (0.01, 0.02, 0.9), False, False), (30000, (0.5, 0.4, 0.3), True, False),
    (9000, (0.004, 0.0, 0.01), False, False), (4097, (1.0, 0.0, 0.0), False, False), (40000, (0.0005, 0.6, 0.0005), True, False),
    (9000, (0.004, 0.002, 0.01), False, True), (5000, (0.03, 0.02, 0.01), True, True), (70000, (0.003, 0.0036, 0.0001), True, True), (3000, (0.08, 0.08, 0.08), True, True)])
def test_batched_compaction_patterns(oracle, n, shares, tail_heavy, same_view):
    """Deletions that fall due at the SECOND, THIRD and FOURTH keyframe of one batched call (stale surfels: ref - lastUpdate > 5 with fewer than five
    updates), in controlled amounts: a handful (the hand-over list), thousands (listed sub-block by sub-block), nearly everything, concentrated at
    the end of the array (relay holes inside the tail), with many or hardly any new surfels to refill the holes.  By default the call is one deferred window behind a classic first keyframe (the map was just uploaded): keyframes with more than 2048
    deletions take the replay's bitmap ordering; under MSL_SF_DEFER=0 (the child-process test above) every keyframe takes the two-kernel chain."""
    from manhattanslam_amd import synth, SURFEL_DTYPE
    g, o = _mk(synth.TUM1)
    rng = np.random.default_rng(n)
    m = synth.surfel_map(n, ref=0, seed=n % 97, min_update_times=5).astype(SURFEL_DTYPE)
    m["pz"] += 100.0                      # out of range: the fusion itself touches nothing, only the stale rule and the compaction act
    m["lastUpdate"] = 9
    u = rng.random(n)
    if tail_heavy:
        u = np.sort(u)[::-1].copy()       # the surfels that go first sit at the end of the array
    lo = 0.0
    for f, sh in enumerate(shares):       # keyframe f + 1 (ref 11 + f) deletes the surfels whose lastUpdate is 5 + f
        sel = (u >= lo) & (u < lo + sh)
        m["lastUpdate"][sel] = 5 + f; m["updateTimes"][sel] = 1 + (f % 3)
        lo += sh
    g.set_batch_capacity(4)
    g.map_reserve(2 * n + 40000)
    g.map_upload(m); o.map_set(m)
    # views 15 degrees apart: every keyframe also spawns > 1000 new surfels (K > D for small D); the same view four times: after the first
    # keyframe hardly any (D > K: leftover holes, tail moves -- with D <= 256 through the register-resident list of the compaction wave)
    frames = [synth.surfel_frame(0 if same_view else 30 * k) for k in range(4)]
    refs = [10, 11, 12, 13]
    g.fuse_resident_batch(refs, np.stack([f[0] for f in frames]), np.stack([f[1] for f in frames]), np.stack([f[2] for f in frames]), [f[3] for f in frames])
    for k, f in enumerate(frames):
        o.fuse_map(refs[k], f[0], f[1], f[2], f[3])
    mo = o.map_get()
    assert_surfels_close(g.map_download(), mo, f"map after the batch ({n}, {shares})")
    c = g.counters()
    assert c["n_live_after"] == len(mo)
    # and the handle keeps working: a second batch on the compacted map
    g.fuse_resident_batch([14, 15, 16], np.stack([f[0] for f in frames[:3]]), np.stack([f[1] for f in frames[:3]]), np.stack([f[2] for f in frames[:3]]), [f[3] for f in frames[:3]])
    for k, f in enumerate(frames[:3]):
        o.fuse_map(14 + k, f[0], f[1], f[2], f[3])
    assert_surfels_close(g.map_download(), o.map_get(), "second batch")
    g.close()


def test_one_gray_upload_for_both_handles(oracle):
    """msl_sf_staged_gray + msl_orb_wait_event (round 6): the ORB extractor reads the gray images a host-image surfel batch staged on the device -- one
    upload for both consumers.  Keypoints and descriptors equal those of the same frames uploaded by the extractor itself, batch after batch (both slot
    sets of the surfel handle, the second batch enqueued while the first one's map stage still runs); strided rows; and the call refuses a handle whose
    last batch had device images."""
    import torch
    from manhattanslam_amd import ORBextractor, SurfelFusion, synth, SURFEL_DTYPE
    from manhattanslam_amd._lib import MslError
    W, H, I = 640, 480, synth.TUM1
    sf = SurfelFusion(W, H, I["fx"], I["fy"], I["cx"], I["cy"], 30.0, 0.5)
    sf.set_batch_capacity(4); sf.map_reserve(200_000)
    sf.map_upload(synth.surfel_map_dense(100_000, ref=0).astype(SURFEL_DTYPE))
    ex = ORBextractor(1000, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=4)
    ref = ORBextractor(1000, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=4)
    cap = ex.capacity
    frames = [synth.surfel_frame(k, variant="B" if k % 3 == 1 else "A") for k in range(8)]
    grays = np.ascontiguousarray(synth.orb_frames(8, w=W, h=H))   # textured images: ~1000 keypoints per frame (the surfel stage takes any gray image)
    depths = np.stack([f[1] for f in frames]); member = frames[0][2]
    for b in range(2):
        sl = slice(4 * b, 4 * b + 4)
        sf.fuse_resident_batch(list(range(4 * b, 4 * b + 4)), grays[sl], depths[sl], member, [frames[j][3] for j in range(4 * b, 4 * b + 4)], member_shared=True)
        p, rs, fs, ev = sf.staged_gray()
        assert rs == W and fs >= W * H
        kps = np.zeros(4 * cap * 28, np.uint8); desc = np.zeros(4 * cap * 32, np.uint8); n = np.zeros(4, np.int32)
        ex.wait_event(ev)
        ex.extract_batch_shared(p, rs, fs, kps, desc, n, 4, W, H)
        kps2 = np.zeros_like(kps); desc2 = np.zeros_like(desc); n2 = np.zeros(4, np.int32)
        ref.extract_batch_host(np.ascontiguousarray(grays[sl]), kps2, desc2, n2, 4, W, H)
        assert n.tolist() == n2.tolist() and n.min() > 100
        for f in range(4):
            assert kps[f * cap * 28:(f * cap + n[f]) * 28].tobytes() == kps2[f * cap * 28:(f * cap + n[f]) * 28].tobytes()
            assert desc[f * cap * 32:(f * cap + n[f]) * 32].tobytes() == desc2[f * cap * 32:(f * cap + n[f]) * 32].tobytes()
    sf.sync()
    # device images: nothing staged
    dg, dd, dm = torch.from_numpy(grays[:4].copy()).cuda(), torch.from_numpy(depths[:4].copy()).cuda(), torch.from_numpy(member.copy()).cuda()
    sf.fuse_resident_batch([8, 9, 10, 11], dg, dd, dm, [frames[j][3] for j in range(4)], device=True, member_shared=True)
    with pytest.raises(MslError):
        sf.staged_gray()
    sf.sync(); sf.close(); ex.close(); ref.close()
