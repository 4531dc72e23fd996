import sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np
from manhattanslam_amd import peac, synth
I = synth.ICL
d = np.stack([synth.depth_u16(synth.surfel_frame(k, intr=I, dropout=0.001)[1]) for k in range(4)])
peac.plane_membership(d, I["fx"], I["fy"], I["cx"], I["cy"], np.float32(1 / 5000.0))
t = time.perf_counter()
m, n = peac.plane_membership(d[:1], I["fx"], I["fy"], I["cx"], I["cy"], np.float32(1 / 5000.0))
print("1 frame ms", (time.perf_counter() - t) * 1e3, n)
