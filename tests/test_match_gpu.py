"""GPU parity: batched Hamming matching by projection (msl_match_by_projection_batch, through the C ABI) vs the CPU oracle's
SearchByProjection restatement.  Integer / index work: the mvpMapPoints index vector and nmatches must be identical."""
import numpy as np
import pytest

from tests import match_scenes as ms

pytestmark = pytest.mark.gpu


def test_descriptor_distance_popcount(oracle):
    from manhattanslam_amd import match
    from tests import oracle_lib
    rng = np.random.default_rng(5)
    a = rng.integers(0, 256, (1000, 32), dtype=np.uint8); b = rng.integers(0, 256, (1000, 32), dtype=np.uint8)
    b[:10] = a[:10]
    assert np.array_equal(match.descriptor_distance(a, b), oracle_lib.descriptor_distance(a, b))


@pytest.mark.parametrize("th,chk", [(15, True), (7, False), (60, True)])
def test_batched_pairs_match_oracle(oracle, th, chk):
    """Eight independent pairs in one call: no motion / forward / backward search modes, sparse and clustered keypoints (the
    clustered ones exceed the 32 stored candidates per point), ragged sizes incl. an empty last frame."""
    from manhattanslam_amd import match, MATCH_PARAMS_DTYPE
    from tests import oracle_lib
    p = ms.params(None, th, chk, dtype=MATCH_PARAMS_DTYPE)
    specs = [(11, 900, 850, 0.0, False), (12, 1016, 1016, 0.3, False), (13, 700, 900, -0.3, False), (14, 1000, 950, 0.3, True),
             (15, 500, 0, 0.0, False), (16, 1, 300, 0.0, False), (17, 1016, 1000, 0.0, True), (18, 800, 800, -0.3, True)]
    cur, last, Tc, Tl = [], [], [], []
    for seed, nc, nl, tz, cluster in specs:
        c, l, a, b = ms.random_pair(seed, p, n_cur=nc, n_last=max(nl, 1), tz=tz, cluster=cluster)
        if nl == 0:
            l = {k: v[:0] for k, v in l.items()}
        cur.append(c); last.append(l); Tc.append(a); Tl.append(b)
    got, nm = match.search_by_projection_batch(p, cur, last, np.stack(Tc), np.stack(Tl))
    tot = 0
    for f in range(len(specs)):
        want, n = oracle_lib.search_by_projection(p, cur[f], last[f], Tc[f], Tl[f])
        assert nm[f] == n, (f, nm[f], n)
        assert np.array_equal(got[f], want), (f, np.flatnonzero(got[f] != want)[:10])
        tot += n
    assert tot > 1500


def test_matching_consumes_orb_frame_outputs(oracle):
    """The pipeline shape: msl_orb_extract_frame_batch outputs of two consecutive frames (frame 1 = frame 0 shifted by a few
    pixels) feed the matcher directly; last-frame map points are the back-projected keypoints of frame 0."""
    from manhattanslam_amd import ORBextractor, frame_params, match, synth, MATCH_PARAMS_DTYPE
    from tests import oracle_lib
    I = synth.TUM1
    img0 = synth.orb_frame(synth.ORB_SEED + 21)
    img1 = np.roll(img0, (3, 5), axis=(0, 1))
    depth = np.full((480, 640), 2.0, np.float32)
    fp = frame_params(I["fx"], I["fy"], I["cx"], I["cy"], 40.0, 640, 480)
    ex = ORBextractor(1000, 1.2, 8, 20, 7, max_batch=2)
    (k0, d0, un0, z0, ur0, c0), (k1, d1, un1, z1, ur1, c1) = ex.extract_frames(np.stack([img0, img1]), np.stack([depth, depth]), fp)
    p = match.match_params(fp, ex.GetScaleFactors(), 15.0, True)
    assert p.dtype == MATCH_PARAMS_DTYPE
    ex.close()
    xyz = np.stack([(un0[:, 0] - I["cx"]) * z0 / I["fx"], (un0[:, 1] - I["cy"]) * z0 / I["fy"], z0], 1).astype(np.float32)   # UnprojectStereo, Twc = I
    rng = np.random.default_rng(3)
    last = dict(xyz=xyz, desc=d0, flags=((z0 > 0).astype(np.uint8) | ((rng.random(len(k0)) < 0.6).astype(np.uint8) << 1)), octave=k0["octave"],
                angle=k0["angle"])
    cur = dict(kps=k1, un_xy=un1, uright=ur1, grid_cell=c1, desc=d1)
    Tl = np.eye(4, dtype=np.float32); Tc = np.eye(4, dtype=np.float32)
    Tc[0, 3] = 5 * 2.0 / I["fx"]; Tc[1, 3] = 3 * 2.0 / I["fy"]
    got, nm = match.search_by_projection_batch(p, [cur], [last], Tc[None], Tl[None])
    want, n = oracle_lib.search_by_projection(p, cur, last, Tc, Tl)
    assert nm[0] == n and np.array_equal(got[0], want)
    assert n > 300      # the shifted frame really is re-found


def test_matcher_handles_own_stream_and_buffers(oracle):
    """msl_match handles (one per ORBmatcher object): two handles used alternately with calls of different sizes (their cached buffers grow
    independently), the host form and the device-resident asynchronous form on the handle's own stream, all equal to the oracle."""
    import torch
    from manhattanslam_amd import match, MATCH_PARAMS_DTYPE
    from manhattanslam_amd.match import Matcher
    from tests import oracle_lib
    p = ms.params(None, 15, True, dtype=MATCH_PARAMS_DTYPE)
    m1, m2 = Matcher(), Matcher()
    for rnd, (nc, nl) in enumerate(((300, 280), (1016, 1000), (120, 500))):
        pairs = [ms.random_pair(200 + 10 * rnd + j, p, n_cur=nc, n_last=nl, tz=0.3 * (j - 1)) for j in range(3)]
        cur = [q[0] for q in pairs]; last = [q[1] for q in pairs]; Tc = np.stack([q[2] for q in pairs]); Tl = np.stack([q[3] for q in pairs])
        for m in (m1, m2):
            got, nm = m.search_by_projection_batch(p, cur, last, Tc, Tl)
            for f in range(3):
                want, n = oracle_lib.search_by_projection(p, cur[f], last[f], Tc[f], Tl[f])
                assert nm[f] == n and np.array_equal(got[f], want), (rnd, f)
    # device-resident form: packed [pairs][cap] arrays as torch tensors, asynchronous until msl_match_sync
    B, cap = 3, 1016
    from manhattanslam_amd._lib import KEYPOINT_DTYPE
    kps = np.zeros((B, cap), KEYPOINT_DTYPE); un = np.zeros((B, cap, 2), np.float32); ur = np.zeros((B, cap), np.float32); cell = np.full((B, cap), -1, np.int32)
    cd = np.zeros((B, cap, 32), np.uint8); ncur = np.zeros(B, np.int32); xyz = np.zeros((B, cap, 3), np.float32); ld = np.zeros((B, cap, 32), np.uint8)
    fl = np.zeros((B, cap), np.uint8); oc = np.zeros((B, cap), np.int32); an = np.zeros((B, cap), np.float32); nlast = np.zeros(B, np.int32)
    pairs = [ms.random_pair(300 + j, p, n_cur=900 + 50 * j, n_last=850, tz=0.0) for j in range(B)]
    for f, (c, l, _, _) in enumerate(pairs):
        n, m = len(c["kps"]), len(l["xyz"])
        ncur[f], nlast[f] = n, m
        kps[f, :n] = c["kps"]; un[f, :n] = c["un_xy"]; ur[f, :n] = c["uright"]; cell[f, :n] = c["grid_cell"]; cd[f, :n] = c["desc"]
        xyz[f, :m] = l["xyz"]; ld[f, :m] = l["desc"]; fl[f, :m] = l["flags"]; oc[f, :m] = l["octave"]; an[f, :m] = l["angle"]
    tc = np.stack([q[2][:3, :4].reshape(12) for q in pairs]).astype(np.float32); tl = np.stack([q[3][:3, :4].reshape(12) for q in pairs]).astype(np.float32)
    dev = [torch.from_numpy(a.view(np.uint8) if a.dtype == KEYPOINT_DTYPE else a).cuda() for a in (kps, un, ur, cell, cd, ncur, xyz, ld, fl, oc, an, nlast, tc, tl)]
    out = torch.full((B, cap), -7, dtype=torch.int32, device="cuda"); nm = torch.zeros(B, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    m1.search_by_projection_device(p, B, cap, dev, out, nm)
    m1.sync()
    out, nm = out.cpu().numpy(), nm.cpu().numpy()
    for f, (c, l, Tc_, Tl_) in enumerate(pairs):
        want, n = oracle_lib.search_by_projection(p, c, l, Tc_, Tl_)
        assert nm[f] == n and np.array_equal(out[f, :len(want)], want), f
    a = np.random.default_rng(1).integers(0, 256, (77, 32), dtype=np.uint8)
    assert np.array_equal(m2.descriptor_distance(a, a[::-1].copy()), oracle_lib.descriptor_distance(a, a[::-1].copy()))
    m1.close(); m2.close()
