"""GPU parity: HIP ORB extractor (through the C ABI) vs the CPU oracle, bit-exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _assert_same(kg, dg, ko, do):
    assert len(kg) == len(ko), (len(kg), len(ko))
    for f in ("x", "y", "size", "angle", "response", "octave", "class_id"):
        a, b = kg[f], ko[f]
        assert np.array_equal(a.view(np.int32) if a.dtype.kind == "f" else a,
                              b.view(np.int32) if b.dtype.kind == "f" else b), f
    if len(kg):
        assert np.array_equal(dg, do)
    else:
        assert dg is None or len(dg) == 0


@pytest.fixture(scope="module")
def gpu_orb():
    from manhattanslam_amd import ORBextractor
    ex = ORBextractor(1000, 1.2, 8, 20, 7, max_width=640, max_height=480, max_batch=8)
    yield ex
    ex.close()


def test_stages_match_oracle(gpu_orb, oracle):
    """Pyramid levels, FAST candidates (order included) and blurred levels are byte-identical."""
    from manhattanslam_amd import synth
    img = synth.orb_frame()
    oex = oracle.orb_create()
    ko, do = oex.extract(img)
    kg, dg = gpu_orb(img)
    for l in range(8):
        if l > 0:
            assert np.array_equal(gpu_orb.debug_level(0, l), oex.level(l)), f"pyramid level {l}"
        assert np.array_equal(gpu_orb.debug_candidates(0, l), oex.candidates(l)), f"candidates level {l}"
        assert np.array_equal(gpu_orb.debug_level(0, l, blurred=True), oex.level(l, blurred=True)), f"blur level {l}"
    _assert_same(kg, dg, ko, do)


@pytest.mark.parametrize("seed_off", [1, 2, 3])
def test_full_extract_bit_exact(gpu_orb, oracle, seed_off):
    from manhattanslam_amd import synth
    img = synth.orb_frame(synth.ORB_SEED + seed_off)
    ko, do = oracle.orb_create().extract(img)
    kg, dg = gpu_orb(img)
    _assert_same(kg, dg, ko, do)
    assert len(kg) >= 1000


def test_batch_matches_single(gpu_orb, oracle):
    from manhattanslam_amd import synth
    imgs = synth.orb_frames(8, seed=77)
    res = gpu_orb.extract_batch(imgs)
    oex = oracle.orb_create()
    for f in range(8):
        ko, do = oex.extract(imgs[f])
        _assert_same(res[f][0], res[f][1], ko, do)


def test_edge_cases(gpu_orb, oracle):
    # constant image: no corners at all -> zero keypoints, descriptors released
    flat = np.full((480, 640), 90, np.uint8)
    k, d = gpu_orb(flat)
    assert len(k) == 0 and d is None
    # empty image: silent return (src/ORBextractor.cc:815-816)
    k, d = gpu_orb(np.zeros((0, 0), np.uint8))
    assert len(k) == 0 and d is None
    # pure noise: far more candidates than quota on every level
    rng = np.random.default_rng(5)
    noise = rng.integers(0, 256, (480, 640), dtype=np.uint8)
    ko, do = oracle.orb_create().extract(noise)
    kg, dg = gpu_orb(noise)
    _assert_same(kg, dg, ko, do)
    # strided input (row stride > width)
    big = np.zeros((480, 700), np.uint8)
    from manhattanslam_amd import synth
    big[:, :640] = synth.orb_frame(1234)
    view = big[:, :640]
    ko, do = oracle.orb_create().extract(np.ascontiguousarray(view))
    kg, dg = gpu_orb(view)
    _assert_same(kg, dg, ko, do)


def test_scale_tables_match_oracle(oracle):
    """a13: the getters of include/ORBextractor.h:58-80 (f32 cumulative products of src/ORBextractor.cc:416-430) and
    mnFeaturesPerLevel (:433-445), bit for bit, for the default and two other parameter sets."""
    from manhattanslam_amd import ORBextractor
    for nf, sc, nl in ((1000, 1.2, 8), (500, 1.2, 6), (2000, 1.5, 5)):
        ex = ORBextractor(nf, sc, nl, 20, 7)
        sf, isf, s2, is2, per, _ = oracle.orb_create(nf, sc, nl, 20, 7).tables()
        for got, want in ((ex.GetScaleFactors(), sf), (ex.GetInverseScaleFactors(), isf), (ex.GetScaleSigmaSquares(), s2),
                          (ex.GetInverseScaleSigmaSquares(), is2)):
            assert np.array_equal(got.view(np.int32), want.view(np.int32))
        assert np.array_equal(ex.features_per_level(), per) and ex.GetLevels() == nl
        assert np.float32(ex.GetScaleFactor()) == np.float32(sc) and ex.capacity == nf + 2 * nl
        ex.close()
    ex = ORBextractor(1000, 1.2, 8, 20, 7)
    assert list(ex.features_per_level()) == [217, 181, 151, 126, 105, 87, 73, 60]
    ex.close()


def test_other_sizes_and_params(oracle):
    """Smaller frame, fewer features/levels, other thresholds."""
    from manhattanslam_amd import ORBextractor, synth
    img = synth.orb_frame(42, 400, 304)
    ex = ORBextractor(500, 1.2, 6, 25, 9, max_width=400, max_height=304)
    ko, do = oracle.orb_create(500, 1.2, 6, 25, 9).extract(img)
    kg, dg = ex(img)
    _assert_same(kg, dg, ko, do)
    ex.close()
    img = synth.orb_frame(43, 1280, 960)
    ex = ORBextractor(1000, 1.2, 8, 20, 7, max_width=1280, max_height=960)
    ko, do = oracle.orb_create().extract(img)
    kg, dg = ex(img)
    _assert_same(kg, dg, ko, do)
    ex.close()


def test_feature_budgets_size_the_quadtree(oracle):
    """The quadtree kernel's node arrays (dynamic LDS) follow the per-level feature budget: a large budget (more than the default dynamic LDS of a
    launch), a budget below four root children, and the default give the oracle's keypoints; low thresholds so that the budgets are reached."""
    from manhattanslam_amd import ORBextractor, synth
    img = synth.orb_frame(77)
    for nf, ini, mn in ((4000, 7, 5), (3000, 20, 7), (40, 20, 7), (24, 20, 7)):
        ex = ORBextractor(nf, 1.2, 8, ini, mn)
        ko, do = oracle.orb_create(nf, 1.2, 8, ini, mn).extract(img)
        kg, dg = ex(img)
        _assert_same(kg, dg, ko, do)
        assert len(kg) > 0
        ex.close()


@pytest.mark.parametrize("w,h,nf,levels,scale", [(640, 160, 137, 4, 1.2), (1280, 200, 1000, 5, 1.2), (640, 480, 60, 12, 1.1), (960, 200, 2000, 3, 1.5),
                                                 (400, 304, 24, 1, 1.2), (1279, 333, 777, 8, 1.2), (1600, 230, 300, 6, 1.2)])
def test_quadtree_node_bound_over_geometries(oracle, w, h, nf, levels, scale):
    """The quadtree's node arrays are sized analytically (max(quota, 4 nIni) + 2 per level): wide images (nIni = round(width / height) up to 8 root
    nodes; tall ones have nIni = 0, which the reference itself cannot handle), tiny and large budgets, 1 to 12 levels, low thresholds so that every
    level reaches its quota.  1600 x 230 with 300 features returns MORE than nfeatures + 2 levels keypoints (the first quadtree round alone makes
    up to 4 nIni nodes per level): the extractor's capacity follows its creation geometry.  An overrun would zero a level (device error); the
    outputs must be the oracle's, bit for bit."""
    from manhattanslam_amd import ORBextractor, synth
    k = max(-(-w // 640), -(-h // 480))
    img = synth.orb_frame(300 + w + nf, 640 * k, 480 * k)[:h, :w].copy()          # the generator makes 4:3 frames: crop one
    ex = ORBextractor(nf, scale, levels, 9, 5, max_width=w, max_height=h)
    ko, do = oracle.orb_create(nf, scale, levels, 9, 5).extract(img)
    kg, dg = ex(img)          # raises on a device-side bound error
    _assert_same(kg, dg, ko, do)
    assert len(kg) >= min(nf, 20) and ex.capacity >= max(len(kg), nf + 2 * levels)
    ex.close()


def test_device_resident_batch(oracle):
    """Asynchronous batch path with inputs and outputs resident in HBM (the bench.py path)."""
    import torch
    from manhattanslam_amd import ORBextractor, synth, KEYPOINT_DTYPE
    B = 5
    imgs = synth.orb_frames(B, seed=901)
    ex = ORBextractor(1000, 1.2, 8, 20, 7, max_batch=B)
    d_img = torch.from_numpy(imgs).cuda()
    cap = ex.capacity
    d_kps = torch.zeros(B * cap * 28, dtype=torch.uint8, device="cuda")
    d_desc = torch.zeros(B * cap * 32, dtype=torch.uint8, device="cuda")
    d_n = torch.zeros(B, dtype=torch.int32, device="cuda")
    ex.extract_batch_device(d_img, d_kps, d_desc, d_n, B, 640, 480)
    ex.sync()
    n = d_n.cpu().numpy()
    kps = d_kps.cpu().numpy().view(KEYPOINT_DTYPE).reshape(B, cap)
    desc = d_desc.cpu().numpy().reshape(B, cap, 32)
    oex = oracle.orb_create()
    for f in range(B):
        ko, do = oex.extract(imgs[f])
        _assert_same(kps[f, :n[f]], desc[f, :n[f]], ko, do)
    ex.close()


@pytest.mark.parametrize("dist", [False, True])
def test_frame_epilogue_matches_oracle(oracle, dist):
    """UndistortKeyPoints + ComputeStereoFromRGBD + AssignFeaturesToGrid fused behind the extraction."""
    from manhattanslam_amd import ORBextractor, frame_params, synth
    from tests import oracle_lib
    I = synth.TUM1
    kw = dict(k1=0.262383, k2=-0.953104, p1=-0.005358, p2=0.002628, k3=1.163314) if dist else {}   # Example/TUM1.yaml:13-17
    pg = frame_params(I["fx"], I["fy"], I["cx"], I["cy"], 40.0, 640, 480, **kw)
    po = oracle_lib.frame_params(I["fx"], I["fy"], I["cx"], I["cy"], 40.0, 640, 480, **kw)
    assert pg.tobytes() == po.tobytes()                       # ComputeImageBounds
    if dist:
        assert pg["minX"][0] != 0 and pg["maxX"][0] != 640
    imgs = synth.orb_frames(3, seed=555)
    depths = np.stack([synth.surfel_frame(k)[1] for k in range(3)])
    ex = ORBextractor(1000, 1.2, 8, 20, 7, max_batch=3)
    res = ex.extract_frames(imgs, depths, pg)
    oex = oracle.orb_create()
    for f in range(3):
        ko, do = oex.extract(imgs[f])
        kg, dg, un, dep, ur, cell = res[f]
        _assert_same(kg, dg, ko, do)
        uo, depo, uro, cello = oracle_lib.frame_epilogue(po, ko, depths[f])
        assert un.tobytes() == uo.tobytes() and dep.tobytes() == depo.tobytes() and ur.tobytes() == uro.tobytes()
        assert np.array_equal(cell, cello)
        assert (dep == -1).sum() > 0 and (dep > 0).sum() > 900 and (cell >= 0).sum() > 900
        if dist:
            assert np.abs(un - np.stack([kg["x"], kg["y"]], 1)).max() > 1.0      # the distortion really moves points
    ex.close()


def test_second_gaussian_generation_variant_library():
    """The product built with MSL_BLUR_VARIANT=1 (manhattanslam_amd/variants/libmsl_blur1.so) against the oracle built the same way:
    blurred levels byte-identical, full extraction bit-exact, and different from the default kernel's descriptors."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "manhattanslam_amd", "variants", "libmsl_blur1.so")
    olib = os.path.join(root, "oracle", "libmsl_oracle_blur1.so")
    assert os.path.exists(lib) and os.path.exists(olib), "python -c 'import __graft_entry__ as g; g.build()' builds both variants"
    script = (
        "import numpy as np, sys\n"
        "sys.path.insert(0, %r)\n"
        "from manhattanslam_amd import ORBextractor, synth\n"
        "from tests import oracle_lib\n"
        "o = oracle_lib.load()\n"
        "assert list(o.gaussian_kernel()) == [18, 34, 48, 56, 48, 34, 18]\n"
        "img = synth.orb_frame(synth.ORB_SEED + 4)\n"
        "ex = ORBextractor(1000, 1.2, 8, 20, 7)\n"
        "kg, dg = ex(img)\n"
        "oe = o.orb_create()\n"
        "ko, do = oe.extract(img)\n"
        "assert kg.tobytes() == ko.tobytes() and np.array_equal(dg, do)\n"
        "for l in (0, 3, 7):\n"
        "    assert np.array_equal(ex.debug_level(0, l, blurred=True), oe.level(l, blurred=True))\n"
        "print('variant ok', len(kg))\n" % root)
    env = dict(os.environ, MSL_LIB=lib, MSL_ORACLE_LIB=olib)
    r = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0 and "variant ok" in r.stdout, r.stdout + r.stderr


def test_pyramid_fused_and_per_level_launches_agree(oracle):
    """k_pyramid (all levels in one launch, tiles chained through LDS) is the default; MSL_ORB_PYRAMID=levels keeps one k_resize launch per
    level (also the fallback when a tile chain does not fit the LDS).  Both give the oracle's levels, at several geometries."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = (
        "import numpy as np, sys\n"
        "sys.path.insert(0, %r)\n"
        "from manhattanslam_amd import ORBextractor, synth\n"
        "from tests import oracle_lib\n"
        "o = oracle_lib.load()\n"
        "for (w, h, nl, sf) in ((640, 480, 8, 1.2), (1280, 960, 8, 1.2), (752, 480, 5, 1.5), (320, 240, 12, 1.1), (480, 360, 3, 2.0)):\n"
        "    img = synth.orb_frame(synth.ORB_SEED + 9, w, h)\n"
        "    ex = ORBextractor(800, sf, nl, 20, 7, max_width=w, max_height=h)\n"
        "    kg, dg = ex(img)\n"
        "    oe = o.orb_create(800, sf, nl, 20, 7)\n"
        "    ko, do = oe.extract(img)\n"
        "    for l in range(1, nl):\n"
        "        assert np.array_equal(ex.debug_level(0, l), oe.level(l)), (w, h, l)\n"
        "    assert kg.tobytes() == ko.tobytes() and np.array_equal(dg, do), (w, h)\n"
        "print('pyramid ok')\n" % root)
    for mode in ("fused", "levels"):
        env = dict(os.environ, MSL_ORB_PYRAMID=mode)
        r = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0 and "pyramid ok" in r.stdout, mode + "\n" + r.stdout + r.stderr
