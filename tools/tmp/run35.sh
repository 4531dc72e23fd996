MSL_PEAC_TIMING=1 python tools/peac_timing.py 2>&1 | grep -v amdgpu | tail -12
python - <<'PY'
import sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from manhattanslam_amd import peac, synth
I = synth.ICL
d = np.stack([synth.depth_u16(synth.surfel_frame(k, intr=I, dropout=0.001)[1]) for k in range(64)])
for thr in (64,):
    peac.plane_membership(d, I["fx"], I["fy"], I["cx"], I["cy"], np.float32(1 / 5000.0))
    t = time.perf_counter()
    for _ in range(3): m, n = peac.plane_membership(d, I["fx"], I["fy"], I["cx"], I["cy"], np.float32(1 / 5000.0))
    print("64 frames ms", (time.perf_counter() - t) * 1e3 / 3, n[:4])
PY
