"""Size-independent properties at BASELINE.json's full size (32-frame batches, ~1 M live surfels, ORB and surfel fusion
running concurrently on their streams as in bench.py): determinism and counter conservation."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run(n_batches=2, F=32, n=1_000_000):
    import torch
    from manhattanslam_amd import ORBextractor, SurfelFusion, synth, SURFEL_DTYPE
    I = synth.TUM1
    orb = ORBextractor(1000, 1.2, 8, 20, 7, max_width=640, max_height=480, max_batch=F)
    sf = SurfelFusion(640, 480, I["fx"], I["fy"], I["cx"], I["cy"], 30.0, 0.5)
    sf.set_batch_capacity(F)
    sf.map_reserve(2 * n)
    sf.map_upload(synth.surfel_map(n, ref=0, min_update_times=1).astype(SURFEL_DTYPE))
    frames = [synth.surfel_frame(k) for k in range(F)]
    gray = torch.from_numpy(np.stack([synth.orb_frame(synth.ORB_SEED + k) for k in range(F)])).cuda()
    depth = torch.from_numpy(np.stack([f[1] for f in frames])).cuda()
    member = torch.from_numpy(frames[0][2]).cuda()
    poses = [f[3] for f in frames]
    cap = orb.capacity
    kps = torch.zeros(F * cap * 28, dtype=torch.uint8, device="cuda")
    desc = torch.zeros(F * cap * 32, dtype=torch.uint8, device="cuda")
    cnt = torch.zeros(F, dtype=torch.int32, device="cuda")
    sizes, counters = [sf.map_size()], []
    ref = 0
    for _ in range(n_batches):
        orb.extract_batch_device(gray, kps, desc, cnt, F, 640, 480)
        sf.fuse_resident_batch(np.arange(ref, ref + F), gray, depth, member, poses, device=True, member_shared=True)
        ref += F
        sf.sync()
        sizes.append(sf.map_size())
        counters.append(sf.counters())
    orb.sync()
    out = (sf.map_download().tobytes(), kps.cpu().numpy().tobytes(), desc.cpu().numpy().tobytes(), cnt.cpu().numpy().copy())
    sf.close()
    return out, sizes, counters


def test_concurrent_front_end_is_deterministic_and_counters_add_up():
    a, sizes, counters = _run()
    b, sizes_b, _ = _run()
    assert a[0] == b[0], "surfel map differs between two identical runs"
    assert a[1] == b[1] and a[2] == b[2] and np.array_equal(a[3], b[3]), "ORB output differs between two identical runs"
    assert sizes == sizes_b
    assert np.all(a[3] > 900) and np.all(a[3] <= 1016)                      # ~1000 features + 2 per level (src/ORBextractor.cc:691-696)
    # the last keyframe of every batch: n_after = n_before - deleted + new, and the map holds exactly n_after surfels
    for c, n_after in zip(counters, sizes[1:]):
        assert c["n_live_after"] == c["n_live_before"] - c["n_deleted"] + c["n_new"] == n_after
        assert 0 <= c["n_updated"] <= c["n_live_before"]
    assert sizes[1] < sizes[0]                                               # young, unconfirmed surfels were removed (:186-192)
