import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from manhattanslam_amd import SurfelFusion, synth, SURFEL_DTYPE
from tests.oracle_lib import OracleSurfel
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
I = synth.TUM1
g = SurfelFusion(640, 480, I['fx'], I['fy'], I['cx'], I['cy'], 30.0, 0.5)
m = synth.surfel_map(N, ref=0).astype(SURFEL_DTYPE)
g.map_reserve(2 * N); g.map_upload(m)
frames = [synth.surfel_frame(k) for k in range(8)]
dev = [(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda(), torch.from_numpy(c).cuda(), p) for a, b, c, p in frames]
# exactness diagnostics on a smaller map
o = OracleSurfel(640, 480, I['fx'], I['fy'], I['cx'], I['cy'], 30.0, 0.5)
g2 = SurfelFusion(640, 480, I['fx'], I['fy'], I['cx'], I['cy'], 30.0, 0.5)
loc = synth.surfel_map(100000, ref=0).astype(SURFEL_DTYPE)
lo, no = o.fuse(2, *frames[2][:3], frames[2][3], loc)
lg = loc.copy(); ng = g2.fuseInitializeMap(2, *frames[2][:3], frames[2][3], lg)
sg, so = g2.debug_seeds(), o.seeds()
for f in ("normX", "posZ", "meanDepth", "viewCos", "size"):
    print("seed", f, "bit-identical:", int((sg[f].view(np.int32) == so[f].view(np.int32)).sum()), "/", len(sg), "maxdiff", float(np.nanmax(np.abs(sg[f] - so[f]))))
for f in ("px", "nx", "size", "weight"):
    print("local", f, "bit-identical:", int((lg[f].view(np.int32) == lo[f].view(np.int32)).sum()), "/", len(lg), "maxdiff", float(np.nanmax(np.abs(lg[f] - lo[f]))))
print("new", len(ng), len(no), "px maxdiff", float(np.abs(ng['px'] - no['px']).max()))
# timing
for k in range(3):
    a, b, c, p = dev[k]; g.fuse_resident(k, a, b, c, p, device=True)
g.sync(); print(g.counters())
K = 20
t = time.time()
for k in range(K):
    a, b, c, p = dev[k % 8]; g.fuse_resident(3 + k, a, b, c, p, device=True)
g.sync(); dt = (time.time() - t) / K
print(f"N={N}: {dt*1e6:.1f} us/keyframe, {1/dt:.0f} kf/s", g.counters())
g.profile_enable(True)
for k in range(K):
    a, b, c, p = dev[k % 8]; g.fuse_resident(30 + k, a, b, c, p, device=True)
for kname, (ms, c) in g.profile_read().items():
    if c: print(f"  {kname:16s} {ms/K*1e3:9.1f} us/keyframe ({c//K} launches)")
