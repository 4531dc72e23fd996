"""Child process of tests/test_surfel_gpu.py::test_default_policy_picks_the_chain_and_keeps_parity: runs WITHOUT MSL_SF_DEFER in the environment, i.e. with
the library's own choice between the classic chain (k_fuse + k_compact per keyframe) and deferred windows (msl_surfel.hip, run_batch):
  a handle with its own two streams                -> classic, always (the shape bench.py runs)
  a handle on ONE caller-provided stream, low churn -> deferred windows behind a classic first keyframe
  the same handle once the churn estimate (spawned + deleted surfels per keyframe, from the asynchronous counter snapshots) exceeds the bound -> classic
Every batch is checked against the oracle.  Exit code 0 = all as stated."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
assert "MSL_SF_DEFER" not in os.environ

import torch  # noqa: E402
from manhattanslam_amd import SurfelFusion, synth, SURFEL_DTYPE  # noqa: E402
from tests.oracle_lib import OracleSurfel  # noqa: E402
from tests.test_surfel_gpu import assert_surfels_close  # noqa: E402

I = synth.TUM1
W, H, B = 640, 480, 8


def run(one_stream, map_kind, n_batches, n_surfels):
    g = SurfelFusion(W, H, I["fx"], I["fy"], I["cx"], I["cy"], 30.0, 0.5)
    o = OracleSurfel(W, H, I["fx"], I["fy"], I["cx"], I["cy"], 30.0, 0.5)
    if one_stream:
        g.set_stream(torch.cuda.current_stream().cuda_stream)
    gray, depth, member, poses, m = synth.bench_inputs(0, B * n_batches, n_surfels, W, H, I, map_kind=map_kind, need_orb_texture=False)
    m = m.astype(SURFEL_DTYPE)
    g.set_batch_capacity(B)
    g.map_reserve(len(m) + 200_000)
    g.map_upload(m); o.map_set(m)
    chain = []
    for c in range(n_batches):
        sl = slice(c * B, c * B + B)
        before = g.debug_scratch(2, which=5).astype(np.int64)
        g.fuse_resident_batch(np.arange(c * B, c * B + B), gray[sl], depth[sl], member, poses[sl], member_shared=True)
        g.sync()                                   # (the counter snapshot of this batch has arrived when the next one is enqueued)
        d = g.debug_scratch(2, which=5).astype(np.int64) - before
        chain.append((int(d[0]), int(d[1])))
        for k in range(c * B, c * B + B):
            o.fuse_map(k, gray[k], depth[k], member, poses[k])
        assert_surfels_close(g.map_download(), o.map_get(), f"{'one stream' if one_stream else 'own streams'}, {map_kind} map, batch {c}")
    g.close()
    return chain


a = run(False, "dense", 3, 120_000)
assert all(d == 0 and cl == B for cl, d in a), ("a handle with its own two streams keeps the classic chain", a)
b = run(True, "dense", 3, 120_000)
assert b[0] == (1, B - 1) and all(cl == 0 and d == B for cl, d in b[1:]), ("one stream, stationary map: deferred windows behind a classic first keyframe", b)
c = run(True, "moving", 6, 150_000)
assert c[0][1] > 0 and any(d == 0 and cl == B for cl, d in c[2:]), ("one stream, hundreds of spawned / deleted surfels per keyframe: the policy falls back to the classic chain", c)
print("policy ok:", a, b, c)
