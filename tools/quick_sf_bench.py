"""Surfel-only micro benchmark for kernel-variant experiments (MSL_LIB selects the library)."""
import sys, os, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from manhattanslam_amd import SurfelFusion, synth, SURFEL_DTYPE
F = 32
I = synth.TUM1
frames = [synth.surfel_frame(f) for f in range(F)]
grays = np.stack([synth.orb_frame(synth.ORB_SEED + f) for f in range(F)]); depths = np.stack([f[1] for f in frames]); member = frames[0][2]
poses = [f[3] for f in frames]
sf = SurfelFusion(640, 480, I["fx"], I["fy"], I["cx"], I["cy"], 30.0, 0.5)
sf.set_batch_capacity(F); sf.map_reserve(2200000)
sf.map_upload(synth.surfel_map(1000000, ref=0, seed=11, min_update_times=5).astype(SURFEL_DTYPE))
dg, dd, dm = torch.from_numpy(grays).cuda(), torch.from_numpy(depths).cuda(), torch.from_numpy(member).cuda()
if os.environ.get("ONE_STREAM"):
    sf.set_stream(torch.cuda.current_stream().cuda_stream)
k = [0]
def step():
    sf.fuse_resident_batch(np.arange(k[0], k[0] + F), dg, dd, dm, poses, device=True, member_shared=True); k[0] += F
for _ in range(4): step()
sf.sync()
names = sf.kernel_names()
sf.profile_enable(1 << names.index("k_fuse"))
t0 = time.perf_counter()
for _ in range(16): step()
sf.sync()
dt = time.perf_counter() - t0
ms, nl = sf.profile_read()["k_fuse"]
print(json.dumps({"lib": os.environ.get("MSL_LIB", "default"), "keyframes_per_s": round(16 * F / dt, 1), "k_fuse_us": round(ms * 1e3 / nl, 2), "ctr": sf.counters()}))
