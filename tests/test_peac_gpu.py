"""GPU parity: the PEAC plane extractor (msl_peac_block_fit / msl_peac_membership_batch through the C ABI) vs the CPU oracle.
FP64 block statistics and PCA byte-identical; the membership image (plane ids, -1, visit counters) identical."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _depth(k, intr, dropout, w=640, h=480):
    from manhattanslam_amd import synth
    _, depth, _, _ = synth.surfel_frame(k, w=w, h=h, intr=intr, dropout=dropout)
    return synth.depth_u16(depth)


def test_block_fit_is_byte_identical(oracle):
    from manhattanslam_amd import peac, synth
    from tests import oracle_lib
    I = synth.ICL
    for k, dr in ((0, 0.0), (100, 0.001), (40, 0.02)):
        d = _depth(k, I, dr)
        d[200:320, 300:420] += 3000                     # a box: depth discontinuities and a second normal direction
        got = peac.block_fit(d, I["fx"], I["fy"], I["cx"], I["cy"], np.float32(1 / 5000.0))[0]
        _, _, want = oracle_lib.peac_run(d, I["fx"], I["fy"], I["cx"], I["cy"], np.float32(1 / 5000.0))
        assert got.tobytes() == want.tobytes(), (k, dr)
        assert (want["nouse"] == 0).sum() > 20


@pytest.mark.parametrize("intr_name", ["ICL", "TUM1"])
def test_membership_matches_oracle_batched(oracle, intr_name):
    """Six frames in one call: clean and noisy depth, one to three walls in view, a box in front, heavy dropout (no plane at all)."""
    from manhattanslam_amd import peac, synth
    from tests import oracle_lib
    I = getattr(synth, intr_name)
    frames = []
    for k, dr, box in ((0, 0.0, False), (40, 0.0005, False), (100, 0.001, True), (170, 0.0, True), (250, 0.003, False), (300, 0.02, False)):
        d = _depth(k, I, dr)
        if box:
            d[150:330, 260:470] = 5000                 # 1 m box front, 90 x 105 cloud points: a plane of its own
        frames.append(d)
    got, n = peac.plane_membership(np.stack(frames), I["fx"], I["fy"], I["cx"], I["cy"], np.float32(1 / 5000.0))
    tot = 0
    for f, d in enumerate(frames):
        want, nw, _ = oracle_lib.peac_run(d, I["fx"], I["fy"], I["cx"], I["cy"], np.float32(1 / 5000.0))
        assert n[f] == nw, (f, n[f], nw)
        assert np.array_equal(got[f], want), (f, np.argwhere(got[f] != want)[:5])
        tot += nw
    assert tot >= 6 and n[5] == 0


def test_other_geometry_and_parameters(oracle):
    """1280x960 depth, 8x6 windows, looser support, INIT_LOOSE, no refinement / segment-border erosion."""
    from manhattanslam_amd import peac, synth
    from tests import oracle_lib
    I = synth.scaled_intrinsics(synth.TUM1, 1280)
    d = _depth(60, I, 0.002, 1280, 960)
    for kw in (dict(window_w=8, window_h=6, min_support=1000), dict(init_loose=1), dict(erode_type=1), dict(do_refine=0), dict(erode_type=0, min_support=500)):
        p = peac.default_params()
        po = oracle_lib.peac_default_params()
        for k, v in kw.items():
            p[k] = v; po[k] = v
        got, n = peac.plane_membership(d, I["fx"], I["fy"], I["cx"], I["cy"], np.float32(1 / 5000.0), params=p)
        want, nw, _ = oracle_lib.peac_run(d, I["fx"], I["fy"], I["cx"], I["cy"], np.float32(1 / 5000.0), params=po)
        assert n[0] == nw and np.array_equal(got[0], want), kw
    assert oracle_lib.peac_default_params().tobytes() == peac.default_params().tobytes()


def test_membership_feeds_surfel_fusion(oracle):
    """Config 4 end to end: the GPU plane extractor's membership image is SurfelFusion's inputPlaneMembershipImg."""
    from manhattanslam_amd import SurfelFusion, peac, synth, SURFEL_DTYPE
    from tests.oracle_lib import OracleSurfel
    from tests.test_surfel_gpu import assert_surfels_close, assert_seeds_close
    I = synth.ICL
    gray, depth, _, pose = synth.surfel_frame(100, intr=I, dropout=0.001)
    member, n = peac.plane_membership(synth.depth_u16(depth), I["fx"], I["fy"], I["cx"], I["cy"], np.float32(1 / 5000.0))
    assert n[0] >= 1 and (member[0] != -1).mean() > 0.5
    g = SurfelFusion(640, 480, I["fx"], I["fy"], I["cx"], I["cy"], 30.0, 0.5)
    o = OracleSurfel(640, 480, I["fx"], I["fy"], I["cx"], I["cy"], 30.0, 0.5)
    local = synth.surfel_map(40000, ref=1).astype(SURFEL_DTYPE)
    lo, no = o.fuse(1, gray, depth, member[0], pose, local)
    lg = local.copy()
    ng = g.fuseInitializeMap(1, gray, depth, member[0], pose, lg)
    assert np.array_equal(g.debug_index(), o.index())
    assert_seeds_close(g.debug_seeds(), o.seeds())
    assert_surfels_close(lg, lo, "local"); assert_surfels_close(ng, no, "new")
    g.close()


def test_device_and_host_clustering_agree(oracle, monkeypatch):
    """The agglomerative clustering runs on the device (k_peac_cluster, one wave per frame) for large calls and on the host workers for small ones
    (and for frames whose node data does not fit the LDS); MSL_PEAC_CLUSTER=device / host forces one side: the same membership image either way."""
    from manhattanslam_amd import peac, synth
    from tests import oracle_lib
    I = synth.ICL
    fac = np.float32(1 / 5000.0)
    frames = []
    for k, dr, box in ((0, 0.0, False), (100, 0.001, True), (170, 0.0, True), (250, 0.003, False), (300, 0.02, False), (33, 0.0005, False), (77, 0.001, True)):
        d = _depth(k, I, dr)
        if box:
            d[150:330, 260:470] = 5000
        frames.append(d)
    stack = np.stack(frames)
    for kw in (dict(), dict(min_support=500, erode_type=0), dict(window_w=8, window_h=6, min_support=1000), dict(max_step=200)):
        p = peac.default_params()
        po = oracle_lib.peac_default_params()
        for k, v in kw.items():
            p[k] = v; po[k] = v
        monkeypatch.setenv("MSL_PEAC_CLUSTER", "device")
        dev, nd = peac.plane_membership(stack, I["fx"], I["fy"], I["cx"], I["cy"], fac, params=p)
        monkeypatch.setenv("MSL_PEAC_CLUSTER", "host")
        host, nh = peac.plane_membership(stack, I["fx"], I["fy"], I["cx"], I["cy"], fac, params=p)
        monkeypatch.delenv("MSL_PEAC_CLUSTER", raising=False)
        assert np.array_equal(nd, nh), (kw, nd, nh)
        assert np.array_equal(dev, host), (kw, np.argwhere(dev != host)[:5])
        for f in (1, 4):
            want, nw, _ = oracle_lib.peac_run(frames[f], I["fx"], I["fy"], I["cx"], I["cy"], fac, params=po)
            assert nd[f] == nw and np.array_equal(dev[f], want), (kw, f)


def test_extract_planes_and_vertex_lists(oracle, monkeypatch):
    """msl_peac_extract_batch = everything PlaneDetection hands on: membership image, extractedPlanes (normal, centre, MSE, N) and plane_vertices_,
    with the clustering on the device and on the host workers."""
    from manhattanslam_amd import peac, synth
    from tests import oracle_lib
    I = synth.ICL
    fac = np.float32(1 / 5000.0)
    frames = []
    for k, dr, box in ((0, 0.0, False), (100, 0.001, True), (170, 0.0, True), (250, 0.003, False), (300, 0.02, False)):
        d = _depth(k, I, dr)
        if box:
            d[150:330, 260:470] = 5000
        frames.append(d)
    ref = []
    for d in frames:
        want, nw, _ = oracle_lib.peac_run(d, I["fx"], I["fy"], I["cx"], I["cy"], fac)
        ref.append((want, nw, oracle_lib.peac_last_planes(want.size)))
    for mode in ("device", "host"):
        monkeypatch.setenv("MSL_PEAC_CLUSTER", mode)
        got, n, planes = peac.extract(np.stack(frames), I["fx"], I["fy"], I["cx"], I["cy"], fac)
        for f, (want, nw, (wp, wv)) in enumerate(ref):
            gp, gv = planes[f]
            assert n[f] == nw and np.array_equal(got[f], want), (mode, f)
            assert gp.tobytes() == wp.tobytes(), (mode, f)
            assert len(gv) == len(wv) and all(np.array_equal(a, b) for a, b in zip(gv, wv)), (mode, f)
    assert sum(r[1] for r in ref) >= 5
    # the organised cloud of PlaneDetection::readDepthImage comes back with the extraction (what adapter/PlaneExtractor.cpp fills cloud.vertices with)
    got, n, planes, cloud = peac.extract(np.stack(frames[:2]), I["fx"], I["fy"], I["cx"], I["cy"], fac, with_cloud=True)
    for f in range(2):
        want_cloud, _ = oracle_lib.peac_block_stats(frames[f], I["fx"], I["fy"], I["cx"], I["cy"], fac)
        assert cloud[f].tobytes() == np.ascontiguousarray(want_cloud).reshape(-1, 3).tobytes(), f
        assert np.array_equal(got[f], ref[f][0])

