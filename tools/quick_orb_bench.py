import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from manhattanslam_amd import ORBextractor, synth, KEYPOINT_DTYPE
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
imgs = synth.orb_frames(min(B, 16))
imgs = np.concatenate([imgs] * ((B + len(imgs) - 1) // len(imgs)))[:B]
ex = ORBextractor(1000, 1.2, 8, 20, 7, max_batch=B)
d_img = torch.from_numpy(imgs).cuda()
d_kps = torch.zeros(B * ex.capacity * 28, dtype=torch.uint8, device='cuda')
d_desc = torch.zeros(B * ex.capacity * 32, dtype=torch.uint8, device='cuda')
d_n = torch.zeros(B, dtype=torch.int32, device='cuda')
for _ in range(3):
    ex.extract_batch_device(d_img, d_kps, d_desc, d_n, B, 640, 480)
ex.sync()
t = time.time(); K = 10
for _ in range(K):
    ex.extract_batch_device(d_img, d_kps, d_desc, d_n, B, 640, 480)
ex.sync()
dt = (time.time() - t) / K
print(f"B={B}: {dt*1e3:.3f} ms/batch, {B/dt:.0f} fps, n={d_n[:4].tolist()}")
ex.profile_enable(-1)
for _ in range(K):
    ex.extract_batch_device(d_img, d_kps, d_desc, d_n, B, 640, 480)
prof = ex.profile_read()
for k, (ms, c) in prof.items():
    if c: print(f"  {k:12s} {ms/K:8.3f} ms/batch  ({c//K} launches)")
