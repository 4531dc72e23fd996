import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from manhattanslam_amd import SurfelFusion, synth, SURFEL_DTYPE
F = 32
I = synth.TUM1
g = SurfelFusion(640, 480, I['fx'], I['fy'], I['cx'], I['cy'], 30.0, 0.5)
g.set_batch_capacity(F)
if len(sys.argv) > 1 and sys.argv[1] == 'serial':
    g.set_stream(torch.cuda.current_stream().cuda_stream)
m = synth.surfel_map(1000000, ref=0, min_update_times=5).astype(SURFEL_DTYPE)
g.map_reserve(2100000); g.map_upload(m)
fr = [synth.surfel_frame(k) for k in range(F)]
gr = torch.from_numpy(np.stack([synth.orb_frame(synth.ORB_SEED + k) for k in range(F)])).cuda()
dp = torch.from_numpy(np.stack([f[1] for f in fr])).cuda(); mb = torch.from_numpy(fr[0][2]).cuda(); poses = [f[3] for f in fr]
ref = 0
for _ in range(3):
    g.fuse_resident_batch(np.arange(ref, ref + F), gr, dp, mb, poses, device=True, member_shared=True); ref += F
g.sync()
t = time.time(); K = 10
for _ in range(K):
    g.fuse_resident_batch(np.arange(ref, ref + F), gr, dp, mb, poses, device=True, member_shared=True); ref += F
g.sync(); dt = (time.time() - t) / K / F
print(f"{dt*1e6:.1f} us/keyframe  {1/dt:.0f} kf/s")
g.profile_enable(-1)
for _ in range(K):
    g.fuse_resident_batch(np.arange(ref, ref + F), gr, dp, mb, poses, device=True, member_shared=True); ref += F
for k, (ms, c) in g.profile_read().items():
    if c: print(f"  {k:16s} {ms/K/F*1e3:8.2f} us/keyframe")
