"""The round-4 synthetic inputs (manhattanslam_amd/synth.py): the furnished room's ray caster and sensor model, and the dense-in-view
live map of SURVEY.md 8(d) config 3.  CPU only; the oracle runs on them so that the scenes stay meaningful parity material."""
import numpy as np

from manhattanslam_amd import synth


def test_dense_map_is_35_percent_in_view_for_every_keyframe_of_a_pass():
    m = synth.surfel_map_dense(300_000)
    fr = [synth.in_view_fraction(m, k) for k in range(0, 64, 7)]
    assert min(fr) > 0.31 and max(fr) < 0.40, fr                       # "~35 % inside the current frustum"
    old = synth.surfel_map(300_000)
    assert synth.in_view_fraction(old, 10) < 0.08                      # the area-uniform map of rounds 1-3: ~6 %
    # creation order: source keyframe, then superpixel raster index -- neighbours in the array are neighbours in some view
    mr = synth.surfel_map_dense(50_000, order="random")
    mc = synth.surfel_map_dense(50_000, order="creation")
    assert abs(synth.in_view_fraction(mr, 5) - synth.in_view_fraction(mc, 5)) < 0.01   # the same distribution, another order
    step = lambda a: float(np.median(np.linalg.norm(np.diff(np.stack([a["px"], a["py"], a["pz"]], 1), axis=0), axis=1)))
    assert step(mc) < 0.5 * step(mr)


def test_ray_caster_known_answers_and_culling():
    sc = dict(spheres=[(np.array([0.0, 0.0, 2.0]), 0.5)], cyls=[(1.5, 2.0, 0.25, -0.5, 1.5)],
              boxes=[(np.array([-1.5, 1.0, 2.0]), np.array([0.3, 0.5, 0.3]), 0.0)])
    o = np.zeros((4, 3))
    d = np.array([[0.0, 0.0, 1.0], [0.75, 0.0, 1.0], [-0.75, 0.5, 1.0], [0.0, -0.5, 1.0]])
    t, n, oid = synth.cast(sc, o, d)
    assert abs(t[0] - 1.5) < 1e-12 and np.allclose(n[0], [0, 0, -1]) and oid[0] == 6                   # sphere front
    assert abs(t[1] - (2.0 - 0.25 / 1.25)) < 1e-12 and oid[1] == 7 and abs(n[1][1]) < 1e-12            # cylinder side: the ray runs through its axis
    assert abs(t[2] - 1.7) < 1e-12 and np.allclose(n[2], [0, 0, -1]) and oid[2] == 8                   # box front face z = 1.7
    assert abs(t[3] - 2.5) < 1e-12 and oid[3] == 4                                                     # past everything: the far wall z = +2.5
    # the frame generator culls solids by their projected bounding sphere: same image as the brute-force cast
    scn = synth.clutter_scene()
    for k in (3, 100):
        gray, depth, member, pose, oid = synth.clutter_frame(k, scene=scn, w=320, h=240, intr=synth.scaled_intrinsics(synth.TUM1, 320))
        T = synth._pose_matrix(k)
        I = synth.scaled_intrinsics(synth.TUM1, 320)
        u, v = np.meshgrid(np.arange(320.0), np.arange(240.0))
        dc = np.stack([(u - I["cx"]) / I["fx"], (v - I["cy"]) / I["fy"], np.ones_like(u)], 2).reshape(-1, 3)
        _, _, o2 = synth.cast(scn, np.broadcast_to(T[:3, 3], dc.shape), dc @ T[:3, :3].T)
        assert np.array_equal(o2.reshape(240, 320), oid)
        assert (oid >= 6).mean() > 0.08 and 0.03 < (depth == 0).mean() < 0.2
        assert member.shape == (120, 160) and (member == -1).all() and pose.shape == (16,) and gray.dtype == np.uint8


def test_furnished_room_exercises_the_rejection_paths_in_the_oracle(oracle):
    """What the bare box room never produced: seeds without depth, seeds whose plane fit is rejected, deletions by occlusion and
    normal disagreement, PEAC leaving non-planar regions free."""
    from tests import oracle_lib
    from tests.oracle_lib import OracleSurfel, SURFEL_DTYPE
    I = synth.TUM1
    sc = synth.clutter_scene()
    o = OracleSurfel(640, 480, I["fx"], I["fy"], I["cx"], I["cy"])
    k = 100
    local = synth.surfel_map_dense(60000, ref=k, scene=sc, k_lo=k - 25, k_hi=k + 35, flip=0.05, floating=0.02, min_update_times=5).astype(SURFEL_DTYPE)
    gray, depth, member, pose, oid = synth.clutter_frame(k, scene=sc, noise_z2=0.004, blobs=0.15)
    lo, no = o.fuse(k, gray, depth, member, pose, local)
    s = o.seeds()
    assert int((s["meanDepth"] == 0).sum()) > 100
    assert int(((s["normX"] == 0) & (s["normY"] == 0) & (s["normZ"] == 0) & (s["use"] == 1)).sum()) > 200
    deleted = (lo["updateTimes"] == 0) & (local["updateTimes"] != 0)
    assert deleted.sum() > 1500                                           # min_update_times = 5: none of them is a stale deletion
    assert (lo["lastUpdate"] == k).sum() > 20000
    mem, npl, _ = oracle_lib.peac_run(synth.depth_u16(depth), I["fx"], I["fy"], I["cx"], I["cy"], np.float32(1 / 5000.0))
    assert npl >= 1 and 0.15 < (mem == -1).mean() < 0.8
