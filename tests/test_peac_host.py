"""Host stage of the PEAC extractor (graph initialisation, agglomerative clustering, erosion, region growing, relabelling) through the C ABI
(msl_peac_membership_from_blocks: no device involved) against the CPU oracle, on the oracle's own block fits.  Runs without a GPU."""
import numpy as np
import pytest


def _depth(k, intr, dropout, w=640, h=480):
    from manhattanslam_amd import synth
    _, depth, _, _ = synth.surfel_frame(k, w=w, h=h, intr=intr, dropout=dropout)
    return synth.depth_u16(depth)


def _scenes(intr):
    frames = []
    for k, dr, box in ((0, 0.0, False), (40, 0.0005, False), (100, 0.001, True), (170, 0.0, True), (250, 0.003, False), (300, 0.02, False)):
        d = _depth(k, intr, dr)
        if box:
            d[150:330, 260:470] = 5000                 # 1 m box front: a plane of its own
        frames.append(d)
    return frames


@pytest.mark.parametrize("intr_name", ["ICL", "TUM1"])
def test_host_stage_matches_oracle(oracle, intr_name):
    from manhattanslam_amd import peac, synth
    from tests import oracle_lib
    I = getattr(synth, intr_name)
    fac = np.float32(1 / 5000.0)
    frames = _scenes(I)
    ref = [oracle_lib.peac_run(d, I["fx"], I["fy"], I["cx"], I["cy"], fac) for d in frames]
    blocks = np.stack([r[2] for r in ref])
    got, n = peac.plane_membership_from_blocks(blocks, np.stack(frames), I["fx"], I["fy"], I["cx"], I["cy"], fac)
    for f, (want, nw, _) in enumerate(ref):
        assert n[f] == nw, (f, n[f], nw)
        assert np.array_equal(got[f], want), (f, np.argwhere(got[f] != want)[:5])
    assert sum(r[1] for r in ref) >= 6 and n[5] == 0


def test_worker_workspaces_are_reusable(oracle):
    """The worker threads keep their workspaces between calls: a frame gives the same image whatever was segmented before it, in calls of
    different sizes, geometries and parameters."""
    from manhattanslam_amd import peac, synth
    from tests import oracle_lib
    I = synth.ICL
    fac = np.float32(1 / 5000.0)
    frames = _scenes(I)
    ref = [oracle_lib.peac_run(d, I["fx"], I["fy"], I["cx"], I["cy"], fac) for d in frames]
    blocks = np.stack([r[2] for r in ref])
    big = _depth(60, synth.scaled_intrinsics(synth.TUM1, 1280), 0.002, 1280, 960)
    Ib = synth.scaled_intrinsics(synth.TUM1, 1280)
    pb = oracle_lib.peac_default_params(); pb["window_w"] = 8; pb["window_h"] = 6; pb["min_support"] = 1000
    wb, nb, bb = oracle_lib.peac_run(big, Ib["fx"], Ib["fy"], Ib["cx"], Ib["cy"], fac, params=pb)
    for order in ([3], [0, 1, 2, 3, 4, 5], [5, 2], [2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2], [4, 3, 1]):
        got, n = peac.plane_membership_from_blocks(blocks[order], np.stack([frames[i] for i in order]), I["fx"], I["fy"], I["cx"], I["cy"], fac)
        for j, i in enumerate(order):
            assert n[j] == ref[i][1] and np.array_equal(got[j], ref[i][0]), (order, j)
        p = peac.default_params(); p["window_w"] = 8; p["window_h"] = 6; p["min_support"] = 1000
        gb, ngb = peac.plane_membership_from_blocks(bb[None], big, Ib["fx"], Ib["fy"], Ib["cx"], Ib["cy"], fac, params=p)
        assert ngb[0] == nb and np.array_equal(gb[0], wb)


def test_host_stage_parameters(oracle):
    from manhattanslam_amd import peac, synth
    from tests import oracle_lib
    I = synth.TUM1
    fac = np.float32(1 / 5000.0)
    d = _depth(100, I, 0.001)
    d[150:330, 260:470] = 5000
    for kw in (dict(init_loose=1), dict(erode_type=1), dict(do_refine=0), dict(erode_type=0, min_support=500), dict(window_w=8, window_h=6, min_support=1000)):
        p = peac.default_params()
        po = oracle_lib.peac_default_params()
        for k, v in kw.items():
            p[k] = v; po[k] = v
        want, nw, blocks = oracle_lib.peac_run(d, I["fx"], I["fy"], I["cx"], I["cy"], fac, params=po)
        got, n = peac.plane_membership_from_blocks(blocks[None], d, I["fx"], I["fy"], I["cx"], I["cy"], fac, params=p)
        assert n[0] == nw and np.array_equal(got[0], want), kw


def test_invalid_arguments_are_rejected():
    from manhattanslam_amd import peac, synth, MslError
    I = synth.TUM1
    p = peac.default_params(); p["min_support"] = 0
    with pytest.raises(MslError):
        peac.plane_membership_from_blocks(np.zeros((1, 768), peac.PEAC_BLOCK_DTYPE), np.zeros((480, 640), np.uint16), I["fx"], I["fy"], I["cx"], I["cy"],
                                          np.float32(1 / 5000.0), params=p)


def test_planes_and_vertex_lists_match_oracle(oracle):
    """msl_peac_extract_from_blocks: plane_filter.extractedPlanes (normal, centre, MSE, N) and plane_vertices_ as PlaneDetection hands them on."""
    from manhattanslam_amd import peac, synth
    from tests import oracle_lib
    I = synth.ICL
    fac = np.float32(1 / 5000.0)
    frames = _scenes(I)
    total = 0
    for kw in (dict(), dict(min_support=500, erode_type=0), dict(erode_type=1)):
        p = peac.default_params(); po = oracle_lib.peac_default_params()
        for k, v in kw.items():
            p[k] = v; po[k] = v
        ref = []
        for d in frames:
            want, nw, blocks = oracle_lib.peac_run(d, I["fx"], I["fy"], I["cx"], I["cy"], fac, params=po)
            ref.append((want, nw, blocks, oracle_lib.peac_last_planes(want.size)))
        got, n, planes = peac.extract_from_blocks(np.stack([r[2] for r in ref]), np.stack(frames), I["fx"], I["fy"], I["cx"], I["cy"], fac, params=p, max_planes=64)
        for f, (want, nw, _, (wp, wv)) in enumerate(ref):
            gp, gv = planes[f]
            assert n[f] == nw and np.array_equal(got[f], want), (kw, f)
            assert gp.tobytes() == wp.tobytes(), (kw, f)
            assert len(gv) == len(wv) and all(np.array_equal(a, b) for a, b in zip(gv, wv)), (kw, f)
            # plane_vertices_ are the pixels that carry the plane's final id, minus stale ids of eroded planes: never more than the image shows
            for j, v in enumerate(gv):
                assert (got[f].ravel()[v] == j).all() and len(v) <= int((got[f] == j).sum())
                total += len(v)
    assert total > 100000


def test_plane_capacity_and_argument_errors(oracle):
    from manhattanslam_amd import peac, synth, MslError
    from tests import oracle_lib
    I = synth.ICL
    fac = np.float32(1 / 5000.0)
    d = _scenes(I)[2]
    want, nw, blocks = oracle_lib.peac_run(d, I["fx"], I["fy"], I["cx"], I["cy"], fac)
    assert nw >= 2
    with pytest.raises(MslError):
        peac.extract_from_blocks(blocks[None], d, I["fx"], I["fy"], I["cx"], I["cy"], fac, max_planes=1)
    p = peac.default_params(); p["do_refine"] = 0
    with pytest.raises(MslError):   # the vertex lists are refineDetails' pMembership
        peac.extract_from_blocks(blocks[None], d, I["fx"], I["fy"], I["cx"], I["cy"], fac, params=p)


def test_worker_count_does_not_change_the_result(oracle):
    """MSL_PEAC_THREADS=1 (everything on the calling thread) and 5 workers give the images the default worker count gives."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = (
        "import sys, hashlib, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "from manhattanslam_amd import peac, synth\n"
        "from tests import oracle_lib\n"
        "from tests.test_peac_host import _scenes\n"
        "I = synth.ICL; fac = np.float32(1 / 5000.0)\n"
        "frames = _scenes(I)\n"
        "blocks = np.stack([oracle_lib.peac_run(d, I['fx'], I['fy'], I['cx'], I['cy'], fac)[2] for d in frames])\n"
        "m, n = peac.plane_membership_from_blocks(blocks, np.stack(frames), I['fx'], I['fy'], I['cx'], I['cy'], fac)\n"
        "print('digest', hashlib.sha256(m.tobytes() + n.tobytes()).hexdigest())\n" % root)
    digests = set()
    for threads in (None, "1", "5"):
        env = dict(os.environ)
        env.pop("MSL_PEAC_THREADS", None)
        if threads:
            env["MSL_PEAC_THREADS"] = threads
        r = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, env=env, timeout=300)
        assert r.returncode == 0 and "digest" in r.stdout, r.stdout + r.stderr
        digests.add(r.stdout.strip().split()[-1])
    assert len(digests) == 1


def test_worker_budget_is_divided_among_local_ranks(oracle):
    """One process per GPU: LOCAL_WORLD_SIZE (torch.distributed.run) divides the CPUs the host stage may use, so 8 ranks on a node do not start
    8 x usable_cpus() workers.  MSL_PEAC_THREADS still overrides."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import numpy as np\n"
            "from manhattanslam_amd import peac, synth\n"
            "from tests import oracle_lib\n"
            "I = synth.ICL; fac = np.float32(1 / 5000.0)\n"
            "d = synth.depth_u16(synth.surfel_frame(3, intr=I, dropout=0.0)[1])\n"
            "_, _, b = oracle_lib.peac_run(d, I['fx'], I['fy'], I['cx'], I['cy'], fac)\n"
            "peac.plane_membership_from_blocks(b[None], d, I['fx'], I['fy'], I['cx'], I['cy'], fac)\n")

    def workers(**envs):
        env = dict(os.environ, MSL_PEAC_POOL_REPORT="1", **envs)
        for k in ("MSL_PEAC_THREADS", "LOCAL_WORLD_SIZE"):
            if k not in envs:
                env.pop(k, None)
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=root, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        line = [ln for ln in r.stderr.splitlines() if "[msl_peac] pool workers" in ln]
        assert line, r.stderr[-2000:]
        return int(line[0].split("=")[1].split()[0]), int(line[0].split("usable CPUs")[1].split(")")[0])

    w1, cpus = workers()
    w8, _ = workers(LOCAL_WORLD_SIZE="8")
    assert w1 == min(64, cpus) and w8 == max(1, cpus // 8)
    assert workers(LOCAL_WORLD_SIZE="8", MSL_PEAC_THREADS="3")[0] == 3


def test_eight_ranks_on_a_shared_cpu_allowance_cluster_on_the_device():
    """BASELINE config 4 on the 8-GPU node: every rank's automatic choice for its 128-keyframe calls.  With the GPU box's allowance (16 usable
    CPUs shared by 8 ranks: 2 workers each) the rule `more than eight frames per worker` sends all ranks to the device path, so the plane
    extractor does not collapse onto shared host cores; a rank with the whole allowance to itself and a small call
    stays on the host, where a lone frame clusters faster (1.8 ms)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = "from manhattanslam_amd._lib import lib\nprint('CHOICE', lib.msl_debug_peac_cluster_on_device(128), lib.msl_debug_peac_cluster_on_device(1))\n"

    def choice(**envs):
        env = {k: v for k, v in os.environ.items() if k not in ("MSL_PEAC_THREADS", "LOCAL_WORLD_SIZE", "MSL_PEAC_CLUSTER")}
        env.update(envs)
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=root, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        a, b = [ln for ln in r.stdout.splitlines() if ln.startswith("CHOICE")][0].split()[1:]
        return int(a), int(b)

    cpus = len(os.sched_getaffinity(0))
    for rank in range(8):      # what torch.distributed.run gives each of the 8 ranks of the node
        assert choice(LOCAL_WORLD_SIZE="8", LOCAL_RANK=str(rank), MSL_PEAC_THREADS="2") == (1, 0)   # 2 workers: 128 > 16 -> device; 1 frame -> host
    if cpus // 8 * 8 < 128:
        assert choice(LOCAL_WORLD_SIZE="8")[0] == 1           # this machine's own CPU count divided by 8 ranks
    assert choice(MSL_PEAC_THREADS="64") == (0, 0)            # one rank with 64 workers: 128 <= 512 -> host
    assert choice(MSL_PEAC_CLUSTER="device", MSL_PEAC_THREADS="64") == (1, 1) and choice(MSL_PEAC_CLUSTER="host", MSL_PEAC_THREADS="1") == (0, 0)


def _stats_of_points(pts):
    from manhattanslam_amd import PEAC_STATS_DTYPE
    s = np.zeros((), PEAC_STATS_DTYPE)
    x, y, z = pts[:, 0], pts[:, 1], pts[:, 2]
    for name, v in (("sx", x), ("sy", y), ("sz", z), ("sxx", x * x), ("syy", y * y), ("szz", z * z), ("sxy", x * y), ("syz", y * z), ("sxz", x * z)):
        s[name] = v.sum()
    s["N"] = len(pts)
    return s


def _mse_cases():
    """Merged-segment statistics of every kind the clustering meets, and the degenerate ones the eigen-solver branches on."""
    from manhattanslam_amd import PEAC_STATS_DTYPE
    rng = np.random.default_rng(20260927)
    out = []
    for i in range(600):   # noisy planes of all orientations, sizes and distances (the sums of one to a few hundred windows)
        n = int(rng.integers(4, 30000))
        nrm = rng.normal(size=3); nrm /= np.linalg.norm(nrm)
        a = np.cross(nrm, [1.0, 0, 0] if abs(nrm[0]) < 0.9 else [0, 1.0, 0]); a /= np.linalg.norm(a)
        b = np.cross(nrm, a)
        uv = rng.uniform(-1, 1, size=(min(n, 400), 2)) * rng.uniform(0.02, 3.0)
        pts = rng.uniform(0.3, 8.0) * nrm + uv[:, :1] * a + uv[:, 1:] * b + rng.normal(size=(len(uv), 1)) * nrm * rng.choice([0.0, 1e-4, 3e-3, 0.05, 0.5])
        s = _stats_of_points(pts)
        if n > len(pts):   # scale the sums up: the same shape with more points
            k = n // len(pts)
            for name in s.dtype.names[:9]:
                s[name] *= k
            s["N"] = len(pts) * k
        out.append(s)
    for axis in range(3):   # exactly axis-aligned planes, lines and single points: zero sub-diagonals, the v1norm2 <= tiny branch, scale == 0
        for kind in ("plane", "line", "point"):
            pts = rng.uniform(-2, 2, size=(50, 3))
            pts[:, axis] = 1.5
            if kind != "plane":
                pts[:, (axis + 1) % 3] = -0.25
            if kind == "point":
                pts[:] = pts[0]
            out.append(_stats_of_points(pts))
    for i in range(300):   # arbitrary symmetric sums: indefinite matrices, huge / tiny magnitudes, equal eigenvalues
        s = np.zeros((), PEAC_STATS_DTYPE)
        mag = 10.0 ** rng.integers(-160, 150)
        for name in s.dtype.names[:9]:
            s[name] = rng.normal() * mag * rng.choice([1.0, 1.0, 1e-8, 0.0])
        s["N"] = int(rng.integers(1, 100000))
        out.append(s)
    for v in (0.0, 1.0, -3.0, 1e-300, 1e300):   # multiples of the identity (td == 0), and N = 0 (sc = inf: NaN everywhere)
        s = np.zeros((), PEAC_STATS_DTYPE); s["sxx"] = s["syy"] = s["szz"] = v; s["N"] = 7
        out.append(s)
        s = s.copy(); s["sxy"] = v * 0.5; out.append(s)
    z = np.zeros((), PEAC_STATS_DTYPE); z["sx"] = 1.0; z["sxx"] = 2.0
    out.append(z)
    return np.array(out, PEAC_STATS_DTYPE)


def test_simd_candidate_mse_is_the_scalar_mse():
    """The clustering evaluates candidate merges several at a time (lock-step lanes, branches as selects): every lane must give the bits the scalar
    eigen-solver gives, whatever shares its group.  Runs on the CPU; widths the CPU cannot execute are skipped, 2 lanes (SSE2) always run."""
    from manhattanslam_amd._lib import lib, ptr
    st = _mse_cases()
    want = np.zeros(len(st))
    assert lib.msl_debug_peac_mse(ptr(st), len(st), 0, ptr(want)) == 0
    assert np.isfinite(want).sum() > 800 and np.isnan(want).any()
    ran = []
    rng = np.random.default_rng(5)
    for lanes in (2, 4, 8, 16):
        got = np.full(len(st), -1.0)
        if lib.msl_debug_peac_mse(ptr(st), len(st), lanes, ptr(got)) != 0:
            continue
        ran.append(lanes)
        same = (got.view(np.uint64) == want.view(np.uint64)) | (np.isnan(got) & np.isnan(want))
        assert same.all(), (lanes, np.flatnonzero(~same)[:5], got[~same][:5], want[~same][:5])
        perm = rng.permutation(len(st))   # other neighbours in the group: same values
        got2 = np.full(len(st), -1.0)
        assert lib.msl_debug_peac_mse(ptr(np.ascontiguousarray(st[perm])), len(st), lanes, ptr(got2)) == 0
        w2 = want[perm]
        assert ((got2.view(np.uint64) == w2.view(np.uint64)) | (np.isnan(got2) & np.isnan(w2))).all(), lanes
    assert 2 in ran
    assert lib.msl_debug_peac_mse(ptr(st), len(st), 3, ptr(want)) != 0


def test_simd_width_does_not_change_the_segmentation(oracle):
    """The whole host stage with the scalar solver (MSL_PEAC_SIMD=0) and with every group width gives the oracle's images."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = (
        "import sys, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "from manhattanslam_amd import peac, synth\n"
        "from tests import oracle_lib\n"
        "from tests.test_peac_host import _scenes\n"
        "I = synth.ICL; fac = np.float32(1 / 5000.0)\n"
        "frames = _scenes(I)\n"
        "ref = [oracle_lib.peac_run(d, I['fx'], I['fy'], I['cx'], I['cy'], fac) for d in frames]\n"
        "m, n = peac.plane_membership_from_blocks(np.stack([r[2] for r in ref]), np.stack(frames), I['fx'], I['fy'], I['cx'], I['cy'], fac)\n"
        "assert all(n[f] == ref[f][1] and np.array_equal(m[f], ref[f][0]) for f in range(len(frames)))\n"
        "print('same as oracle')\n" % root)
    for envs in ({"MSL_PEAC_SIMD": "0"}, {"MSL_PEAC_SIMD": "2"}, {"MSL_PEAC_LANES": "4"}, {"MSL_PEAC_LANES": "8"}, {"MSL_PEAC_SIMD": "4"}, {}):
        env = dict(os.environ)
        env.pop("MSL_PEAC_SIMD", None); env.pop("MSL_PEAC_LANES", None)
        env.update(envs)
        r = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, env=env, timeout=300)
        assert r.returncode == 0 and "same as oracle" in r.stdout, (envs, r.stdout + r.stderr[-2000:])
