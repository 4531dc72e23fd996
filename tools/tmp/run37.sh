python - <<'PY' 2>&1 | grep -v "graph\|amdgpu"
import sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from manhattanslam_amd import peac, synth
I = synth.ICL
d = np.stack([synth.depth_u16(synth.surfel_frame(k, intr=I, dropout=0.001)[1]) for k in range(64)])
peac.plane_membership(d, I["fx"], I["fy"], I["cx"], I["cy"], np.float32(1 / 5000.0))
os.environ["MSL_PEAC_TIMING"] = "1"
for n in (1, 8, 16, 32, 64):
    t = time.perf_counter()
    m, npl = peac.plane_membership(d[:n], I["fx"], I["fy"], I["cx"], I["cy"], np.float32(1 / 5000.0))
    print(n, "frames ms", (time.perf_counter() - t) * 1e3)
PY
