"""Floors: superpixel stage alone (tiny map), at several batch sizes."""
import sys, os, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from manhattanslam_amd import SurfelFusion, synth, SURFEL_DTYPE
I = synth.TUM1
NMAP = int(os.environ.get("NMAP", "1000"))
for F in (16, 32, 64):
    frames = [synth.surfel_frame(f) for f in range(F)]
    grays = np.stack([synth.orb_frame(synth.ORB_SEED + f) for f in range(F)]); depths = np.stack([f[1] for f in frames]); member = frames[0][2]
    poses = [f[3] for f in frames]
    sf = SurfelFusion(640, 480, I["fx"], I["fy"], I["cx"], I["cy"], 30.0, 0.5)
    sf.set_batch_capacity(F); sf.map_reserve(2200000)
    sf.map_upload(synth.surfel_map(NMAP, ref=0, seed=11, min_update_times=5).astype(SURFEL_DTYPE))
    dg, dd, dm = torch.from_numpy(grays).cuda(), torch.from_numpy(depths).cuda(), torch.from_numpy(member).cuda()
    k = [0]
    def step():
        sf.fuse_resident_batch(np.arange(k[0], k[0] + F), dg, dd, dm, poses, device=True, member_shared=True); k[0] += F
    for _ in range(3): step()
    sf.sync()
    R = 512 // F
    t0 = time.perf_counter()
    for _ in range(R): step()
    sf.sync()
    dt = time.perf_counter() - t0
    print(json.dumps({"nmap": NMAP, "batch": F, "us_per_keyframe": round(dt / (R * F) * 1e6, 2), "n": sf.counters()["n_live_after"]}))
    del sf
