"""GPU tests of the screen-position dealing of k_fuse's sub-blocks to the eight XCDs (round 6; msl_sf_map.hip: deal_subblocks).

The dealing is a HINT for speed -- workgroup g runs on XCD g % 8, and an XCD whose waves all project into one band of image rows fetches that band
of the texel map and of the seed records instead of the whole screen -- but the table must be a permutation of the grid whatever the keys are,
or a sub-block would be fused twice or not at all (the parity tests of test_surfel_gpu.py / test_clutter_gpu.py run with the dealing on)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def deal(keys):
    import ctypes as C
    from manhattanslam_amd._lib import lib, check
    keys = np.ascontiguousarray(keys, np.uint32)
    out = np.zeros(len(keys), np.uint32)
    check(lib.msl_debug_deal(keys.ctypes.data_as(C.c_void_p), len(keys), out.ctypes.data_as(C.c_void_p)))
    return out


def check_deal(keys, table):
    G = len(keys)
    gs = G // 8
    assert np.array_equal(np.sort(table), np.arange(G, dtype=np.uint32)), "not a permutation of the grid"
    k = np.minimum(keys, 255)[table].reshape(8, gs)          # row x: the keys of XCD x's waves in dispatch order
    inview = k < 255
    n_in = inview.sum(axis=1)
    NI = int((np.minimum(keys, 255) < 255).sum())
    assert n_in.sum() == NI
    assert n_in.max() - n_in.min() <= 1, ("in-view sub-blocks per XCD", n_in)                  # equal shares whatever the distribution of the rows
    for x in range(8):
        assert inview[x, :n_in[x]].all() and not inview[x, n_in[x]:].any(), "the in-view sub-blocks are dispatched first"
        assert np.all(np.diff(k[x, :n_in[x]].astype(int)) >= 0), "top to bottom inside an XCD"
    bands = [(k[x, :n_in[x]].min(), k[x, :n_in[x]].max()) for x in range(8) if n_in[x]]
    for (lo0, hi0), (lo1, hi1) in zip(bands, bands[1:]):
        assert hi0 <= lo1, ("bands of image rows must not interleave", bands)


@pytest.mark.parametrize("G", [8, 64, 7808, 8192, 8200, 62528])
@pytest.mark.parametrize("kind", ["uniform", "none", "all", "one_row", "clustered", "unwritten"])
def test_dealing_is_a_balanced_banded_permutation(G, kind):
    rng = np.random.default_rng(G * 7 + len(kind))
    if kind == "uniform":
        keys = np.where(rng.random(G) < 0.7, rng.integers(0, 255, G), 255)
    elif kind == "none":
        keys = np.full(G, 255)
    elif kind == "all":
        keys = rng.integers(0, 255, G)
    elif kind == "one_row":
        keys = np.where(rng.random(G) < 0.9, 17, 255)              # one bin holds everything: the bin is cut by rank, not as a whole
    elif kind == "clustered":
        keys = np.where(rng.random(G) < 0.4, np.clip(rng.normal(200, 6, G), 0, 254).astype(int), 255)
    else:
        keys = np.where(rng.random(G) < 0.5, 0xFFFFFFFF, rng.integers(0, 300, G))                # never-written entries and out-of-range values count as 255
    keys = keys.astype(np.uint32)
    check_deal(keys, deal(keys))


def test_rejects_grids_that_are_no_multiple_of_eight():
    import ctypes as C
    from manhattanslam_amd._lib import lib
    k = np.zeros(12, np.uint32)
    o = np.zeros(12, np.uint32)
    assert lib.msl_debug_deal(k.ctypes.data_as(C.c_void_p), 12, o.ctypes.data_as(C.c_void_p)) != 0


def test_resident_batches_run_dealt_and_keys_follow_the_screen(oracle):
    """After a resident batch on the dense map the handle holds a table for its grid; the keys k_fuse left are the mean image row of each sub-block's
    in-view surfels (checked against a projection of the uploaded map with the last keyframe's pose), and the table the next launch would use
    deals them into bands."""
    from manhattanslam_amd import SurfelFusion, synth, SURFEL_DTYPE
    I = synth.TUM1
    W, H, F, n = 640, 480, 4, 200_000
    g = SurfelFusion(W, H, I["fx"], I["fy"], I["cx"], I["cy"], 30.0, 0.5)
    m = synth.surfel_map_dense(n, ref=0).astype(SURFEL_DTYPE)
    g.set_batch_capacity(F)
    g.map_reserve(n + 100_000)
    g.map_upload(m)
    frames = [synth.surfel_frame(k) for k in range(F)]
    for rep in range(2):
        g.fuse_resident_batch(np.arange(rep * F, rep * F + F), np.stack([f[0] for f in frames]), np.stack([f[1] for f in frames]), frames[0][2], [f[3] for f in frames],
                              member_shared=True)
    G = int(g.debug_scratch(1, which=4)[0])
    assert G >= (n + 127) // 128 and G % 8 == 0, G
    keys, table = g.debug_scratch(G, which=2), g.debug_scratch(G, which=3)
    import os
    if os.environ.get("MSL_SF_DEFER") == "0":
        # classic chain: the table is rebuilt on every 4th keyframe of a call (here: keyframe 4) while the keys are those of the LAST launch (keyframe 7) --
        # a table a few keyframes old is still a permutation of the grid, but not the exact dealing of the newest keys
        assert np.array_equal(np.sort(table), np.arange(G, dtype=np.uint32)), "not a permutation of the grid"
    else:
        check_deal(keys, table)   # deferred chain (the suite's default): one dealing per window, from the keys of its last keyframe
    nsb = (g.map_size() + 127) // 128
    assert (keys[:nsb] < 255).sum() > 0.3 * nsb and np.all(keys[nsb + 1:G] >= 255)
    # the keys against a float64 projection of the downloaded map with the last keyframe's pose (the map moved a little since: compare the medians)
    mm = g.map_download()
    T = np.linalg.inv(np.asarray(frames[F - 1][3], np.float64).reshape(4, 4).T)     # column-major pose -> camera from world
    p = np.stack([mm["px"], mm["py"], mm["pz"], np.ones(len(mm))], axis=1) @ T.T
    z = p[:, 2]
    with np.errstate(divide="ignore", invalid="ignore"):
        u, v = p[:, 0] * I["fx"] / z + I["cx"], p[:, 1] * I["fy"] / z + I["cy"]
    vis = (z > 0.5) & (z < 30.0) & (u >= 0.5) & (u < W - 1.5) & (v >= 0.5) & (v < H - 1.5) & (mm["updateTimes"] > 0)
    err = []
    for sb in range(0, nsb - 1, 7):
        s = slice(sb * 128, sb * 128 + 128)
        if vis[s].sum() >= 16 and keys[sb] < 255:
            err.append(abs(float(keys[sb]) - v[s][vis[s]].mean() * 254.0 / H))
    assert len(err) > 50 and np.median(err) < 3.0, (len(err), np.median(err) if err else None)
    g.close()
