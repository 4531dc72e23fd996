"""Per-wave timeline of k_fuse inside the frontend workload (library built with -DMSL_FUSE_STAMPS; MSL_LIB selects it):
start / after-gather / end stamps (100 MHz device clock) of every wave of the LAST keyframe of a pass, with ORB and the batched
superpixel stage running next to it as in bench.py.  Prints the distribution of wave start delays and lifetimes."""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from manhattanslam_amd import SurfelFusion, ORBextractor, synth, SURFEL_DTYPE
F, B = 256, 32
alone = bool(os.environ.get("ALONE"))
I = synth.TUM1
frames = [synth.surfel_frame(f) for f in range(64)]
grays = np.stack([synth.orb_frame(synth.ORB_SEED + f) for f in range(64)]); depths = np.stack([f[1] for f in frames]); member = frames[0][2]
poses = [f[3] for f in frames]
sf = SurfelFusion(640, 480, I["fx"], I["fy"], I["cx"], I["cy"], 30.0, 0.5)
sf.set_batch_capacity(B); sf.map_reserve(2200000)
dense = os.environ.get("MAP", "dense")
smap = synth.surfel_map(1000000, ref=0, seed=11, min_update_times=5) if dense == "sparse" else synth.surfel_map_dense(1000000, order=os.environ.get("ORDER", "creation"))
sf.map_upload(smap.astype(SURFEL_DTYPE)); sf.map_snapshot()
orb = ORBextractor(1000, 1.2, 8, 20, 7, max_batch=B)
dg = torch.from_numpy(grays).cuda().repeat(F // 64, 1, 1).contiguous(); dd = torch.from_numpy(depths).cuda().repeat(F // 64, 1, 1).contiguous(); dm = torch.from_numpy(member).cuda()
cap = orb.capacity
dk = torch.zeros(F * cap * 28, dtype=torch.uint8, device="cuda"); ds = torch.zeros(F * cap * 32, dtype=torch.uint8, device="cuda"); dn = torch.zeros(F, dtype=torch.int32, device="cuda")
if alone:
    sf.set_stream(torch.cuda.current_stream().cuda_stream)
def one_pass():
    sf.map_restore()
    for sb in range(F // B):
        if not alone and not os.environ.get("NOORB"):
            orb.extract_batch_device(dg[sb * B:], dk[sb * B * cap * 28:], ds[sb * B * cap * 32:], dn[sb * B:], B, 640, 480)
        sf.fuse_resident_batch(np.arange(sb * B, sb * B + B), dg[sb * B:], dd[sb * B:], dm, [poses[(sb * B + j) % 64] for j in range(B)], device=True, member_shared=True)
for _ in range(3): one_pass()
sf.sync(); orb.sync()
n = sf.map_size()
nsub = ((n + 255) // 256 + int(os.environ.get("FUSE_SB", "2")) - 1) // int(os.environ.get("FUSE_SB", "2"))   # waves
w = sf.debug_scratch(8 * (nsub + 1024)).reshape(-1, 8)
cst = sf.debug_scratch(8, which=1).astype(np.int64)
sec = sf.debug_scratch(20, which=1, offset=64).astype(np.float64)
if sec[0] > 0:
    names = ["kernarg+frame record", "seed record", "window loads+ownership", "ordered lists", "positions+normals", "inliers", "six sums", "GN1", "GN2", "GN3", "GN4", "GN5"]
    print("kb_seed_plane sections (shader cycles per wave, slot 0, %d waves): " % sec[0] + ", ".join(f"{n} {sec[1 + i] / sec[0]:.0f}" for i, n in enumerate(names)))
usec = sf.debug_scratch(8, which=1, offset=96).astype(np.float64)
if usec[0] > 0:
    names = ["seed record + gather + depth list", "means + colour + depth sum", "Newton steps", "stores"]
    print("kb_update_seeds sections (shader cycles per wave, slot 0, all three passes, %d waves): " % usec[0] + ", ".join(f"{n} {usec[1 + i] / usec[0]:.0f}" for i, n in enumerate(names)))
w = w[(w[:, 0] != 0) & (w[:, 0].astype(np.int64) > w[:, 0].astype(np.int64).max() - 100000)]   # waves of the stamped launch only (keyframe MSL_FUSE_STAMPS of the last pass)
nsub = len(w)
t0, t1, t2, tot = (w[:, i].astype(np.int64) for i in range(4))
base = t0.min()
start = (t0 - base) * 0.01; gather = (t1 - t0) * 0.01; life = (t2 - t0) * 0.01; end = (t2 - base) * 0.01
xcc = w[:, 5] & 15
os.makedirs("gpurun_out", exist_ok=True)
np.save(f"gpurun_out/fuse_stamps_{'alone' if alone else ('noorb' if os.environ.get('NOORB') else 'region')}.npy", w)
pc = lambda a: [round(float(np.percentile(a, q)), 2) for q in (5, 25, 50, 75, 95, 100)]
print("k_compact workgroup 0 (us): loads %.2f, scans %.2f, emission %.2f, tail %.2f; after k_fuse's first wave start: start %.2f end %.2f; K=%d D=%d" % (
    (cst[1] - cst[0]) * 0.01, (cst[2] - cst[1]) * 0.01, (cst[3] - cst[2]) * 0.01, (cst[4] - cst[3]) * 0.01, (cst[0] - base) * 0.01, (cst[4] - base) * 0.01, cst[5], cst[6]))
dense_w = w[:, 7] == 1
order = np.argsort(w[:, 0].astype(np.int64))
print("per-decile of wave START order: mean lifetime us / mean survivors / dense share:", [(round(float(life[o].mean()), 1), int(tot[o].mean()), round(float(dense_w[o].mean()), 2)) for o in np.array_split(order, 10)])
print("lifetime by survivors bucket (0, 1-63, 64-127, 128-191, 192-256):", [round(float(life[(tot >= a) & (tot <= b)].mean()), 2) if ((tot >= a) & (tot <= b)).any() else None for a, b in ((0, 0), (1, 63), (64, 127), (128, 191), (192, 256))],
      "counts", [int(((tot >= a) & (tot <= b)).sum()) for a, b in ((0, 0), (1, 63), (64, 127), (128, 191), (192, 256))])
print("phase A by survivors bucket:", [round(float(gather[(tot >= a) & (tot <= b)].mean()), 2) if ((tot >= a) & (tot <= b)).any() else None for a, b in ((0, 0), (1, 63), (64, 127), (128, 191), (192, 256))])
print(json.dumps({"lib": os.environ.get("MSL_LIB", "default"), "alone": alone, "map": dense, "waves": int(nsub), "kernel_span_us": round(float(end.max()), 2),
                  "start_delay_us_pct": pc(start), "phaseA_us_pct": pc(gather), "lifetime_us_pct": pc(life), "survivors_pct": pc(tot),
                  "waves_per_xcc": np.bincount(xcc, minlength=8).tolist(), "pipe_ids": np.bincount((w[:, 4] >> 6) & 3, minlength=4).tolist(), "queue_ids": np.bincount((w[:, 4] >> 24) & 7, minlength=8).tolist(),
                  "slot_ids": np.bincount(w[:, 4] & 15, minlength=8).tolist(),
                  "concurrency_at_us": {str(t): int(((start <= t) & (end > t)).sum()) for t in (1, 2, 4, 6, 8, 10, 12, 14, 16, 18)}}))
