"""Single-frame latency of the PEAC drop-in call (device block fit + host clustering), the reference's call pattern."""
import sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from manhattanslam_amd import peac, synth
I = synth.ICL
W, H = 640, 480
d = np.stack([synth.depth_u16(synth.surfel_frame(k, intr=I, dropout=0.001)[1]) for k in range(8)])
d16 = torch.from_numpy(d.view(np.int16)).cuda().contiguous()
prm = peac.default_params()
hm = np.zeros((1, H // 2, W // 2), np.int32); hn = np.zeros(1, np.int32)
def call(k):
    t = time.perf_counter()
    peac.plane_membership_device(d16[k:k + 1], 1, 1, W, H, I["fx"], I["fy"], I["cx"], I["cy"], np.float32(1 / 5000.0), prm, hm, hn)
    return (time.perf_counter() - t) * 1e3
for k in range(8): call(k)
ts = sorted(call(k % 8) for k in range(64))
print("device-resident depth: single frame ms min %.2f med %.2f p90 %.2f" % (ts[0], ts[32], ts[57]), "SIMD", os.environ.get("MSL_PEAC_SIMD"), "LANES", os.environ.get("MSL_PEAC_LANES"))
# host depth in, host membership out (adapter/PlaneExtractor.cpp's shape)
def call_host(k):
    t = time.perf_counter()
    peac.plane_membership(d[k], I["fx"], I["fy"], I["cx"], I["cy"], np.float32(1 / 5000.0))
    return (time.perf_counter() - t) * 1e3
try:
    for k in range(8): call_host(k)
    ts = sorted(call_host(k % 8) for k in range(64))
    print("host depth:            single frame ms min %.2f med %.2f p90 %.2f" % (ts[0], ts[32], ts[57]))
except Exception as e:
    print("host call failed", e)
