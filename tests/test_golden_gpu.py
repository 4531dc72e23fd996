"""GPU vs the committed golden vectors (tests/golden/*.npz) -- independent of the oracle library at run time."""
import hashlib
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_orb_golden_bit_exact():
    from manhattanslam_amd import ORBextractor, synth
    g = np.load(os.path.join(GOLD, "orb_640x480.npz"))
    ex = ORBextractor(1000, 1.2, 8, 20, 7)
    k, d = ex(synth.orb_frame(int(g["seed"])))
    assert k.tobytes() == g["keypoints"].tobytes() and np.array_equal(d, g["descriptors"])
    ex.close()
    g = np.load(os.path.join(GOLD, "orb_400x304.npz"))
    ex = ORBextractor(500, 1.2, 6, 25, 9, max_width=400, max_height=304)
    k, d = ex(synth.orb_frame(int(g["seed"]), 400, 304))
    assert k.tobytes() == g["keypoints"].tobytes() and np.array_equal(d, g["descriptors"])
    ex.close()


def test_surfel_golden_within_tolerance():
    from manhattanslam_amd import SurfelFusion, synth, SURFEL_DTYPE
    from tests.test_surfel_gpu import assert_surfels_close
    g = np.load(os.path.join(GOLD, "surfel_640x480_B.npz"))
    k = int(g["frame"])
    gray, depth, member, pose = synth.surfel_frame(k, variant="B")
    local = synth.surfel_map(int(g["n_local"]), ref=k).astype(SURFEL_DTYPE)
    I = synth.TUM1
    sf = SurfelFusion(640, 480, I["fx"], I["fy"], I["cx"], I["cy"], 30.0, 0.5)
    before = local.copy()
    new = sf.fuseInitializeMap(k, gray, depth, member, pose, local)
    assert_surfels_close(new, g["new_surfels"].view(SURFEL_DTYPE), "new")
    ci = g["changed_index"]
    assert_surfels_close(local[ci], g["changed_surfels"].view(SURFEL_DTYPE), "changed")
    untouched = np.setdiff1d(np.arange(len(local)), ci)
    assert local[untouched].tobytes() == before[untouched].tobytes()
    sf.close()


def test_peac_golden_bit_exact():
    from manhattanslam_amd import synth, peac
    g = np.load(os.path.join(GOLD, "peac_640x480.npz"))
    _, depth, _, _ = synth.surfel_frame(int(g["frame"]), variant="B")
    d16 = np.clip(np.round(depth * 5000.0), 0, 65535).astype(np.uint16)
    d16[200:320, 300:420] += 4000
    assert hashlib.sha256(d16.tobytes()).hexdigest() == str(g["depth16_sha256"])
    I = synth.TUM1
    cloud, st = peac.block_stats(d16, I["fx"], I["fy"], I["cx"], I["cy"], np.float32(1.0 / 5000.0))
    assert hashlib.sha256(cloud[0].tobytes()).hexdigest() == str(g["cloud_sha256"])
    assert st[0].tobytes() == g["stats"].tobytes()
    assert 0 < int((st[0]["nouse"] == 1).sum()) < st.shape[1]


def test_peac_membership_golden_bit_exact(monkeypatch):
    """The whole plane extractor (block fit and clustering on the device, pixel stages on the host) against the committed membership image."""
    from manhattanslam_amd import synth, peac
    monkeypatch.setenv("MSL_PEAC_CLUSTER", "device")
    g = np.load(os.path.join(GOLD, "peac_membership_640x480.npz"))
    I = synth.ICL
    _, depth, _, _ = synth.surfel_frame(int(g["frame"]), intr=I, dropout=float(g["dropout"]))
    d16 = synth.depth_u16(depth)
    member, n = peac.plane_membership(np.stack([d16, d16, d16]), I["fx"], I["fy"], I["cx"], I["cy"], np.float32(1.0 / 5000.0))
    for f in range(3):
        assert n[f] == int(g["nplanes"]) and np.array_equal(member[f], g["membership"]), f
    assert peac.block_fit(d16, I["fx"], I["fy"], I["cx"], I["cy"], np.float32(1.0 / 5000.0))[0].tobytes() == g["blocks"].tobytes()


def test_match_golden_bit_exact():
    from manhattanslam_amd import match, MATCH_PARAMS_DTYPE
    from tests import match_scenes as ms
    g = np.load(os.path.join(GOLD, "match_pairs.npz"))
    p = ms.params(None, float(g["th"]), True, dtype=MATCH_PARAMS_DTYPE)
    cur, last, Tc, Tl = [], [], [], []
    for name in ("a", "b"):
        seed, nc, nl, cluster = (int(v) for v in g[f"{name}_spec"])
        c, l, a, b = ms.random_pair(seed, p, n_cur=nc, n_last=nl, tz=float(g[f"{name}_tz"]), cluster=bool(cluster))
        assert hashlib.sha256(c["desc"].tobytes() + l["desc"].tobytes() + l["xyz"].tobytes()).hexdigest() == str(g[f"{name}_input_sha256"])
        cur.append(c); last.append(l); Tc.append(a); Tl.append(b)
    got, nm = match.search_by_projection_batch(p, cur, last, np.stack(Tc), np.stack(Tl))
    for f, name in enumerate(("a", "b")):
        assert nm[f] == int(g[f"{name}_nmatches"]) and np.array_equal(got[f], g[f"{name}_matches"]), name


def test_clutter_golden_within_tolerance():
    """The furnished-room golden vector (tests/golden/clutter_640x480.npz): SurfelFusion within 1e-4, the plane extractor's membership identical."""
    from manhattanslam_amd import SurfelFusion, peac, synth, SURFEL_DTYPE
    from tests.test_surfel_gpu import assert_surfels_close, assert_seeds_close
    g = np.load(os.path.join(GOLD, "clutter_640x480.npz"))
    k = int(g["frame"])
    sc = synth.clutter_scene()
    gray, depth, member, pose, _ = synth.clutter_frame(k, scene=sc)
    assert hashlib.sha256(depth.tobytes()).hexdigest() == str(g["depth_sha256"])
    local = synth.surfel_map_dense(int(g["n_local"]), ref=k, scene=sc, k_lo=k - 25, k_hi=k + 35, flip=0.05, floating=0.02, min_update_times=1).astype(SURFEL_DTYPE)
    I = synth.TUM1
    sf = SurfelFusion(640, 480, I["fx"], I["fy"], I["cx"], I["cy"], 30.0, 0.5)
    before = local.copy()
    new = sf.fuseInitializeMap(k, gray, depth, member, pose, local)
    assert_surfels_close(new, g["new_surfels"].view(SURFEL_DTYPE), "new")
    ci = g["changed_index"]
    assert_surfels_close(local[ci], g["changed_surfels"].view(SURFEL_DTYPE), "changed")
    untouched = np.setdiff1d(np.arange(len(local)), ci)
    assert local[untouched].tobytes() == before[untouched].tobytes()
    from tests.oracle_lib import SEED_DTYPE
    assert_seeds_close(sf.debug_seeds(), g["seeds"].view(SEED_DTYPE))
    assert hashlib.sha256(sf.debug_index().tobytes()).hexdigest() == str(g["index_sha256"])
    sf.close()
    mem, n = peac.plane_membership(synth.depth_u16(depth), I["fx"], I["fy"], I["cx"], I["cy"], np.float32(1.0 / 5000.0))
    assert n[0] == int(g["peac_nplanes"]) and np.array_equal(mem[0], g["peac_membership"].astype(np.int32))
