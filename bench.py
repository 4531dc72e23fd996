#!/usr/bin/env python3
"""Benchmark of the MI355X-native RGB-D front end (ORB extractor + surfel fusion).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One *step* = one batch of F synthetic 640x480 RGB-D frames through the whole front end on each GPU:
ORB extraction of the F gray frames (one frame-batched launch sequence) and, frame after frame, surfel
fusion of every frame (every frame is treated as a keyframe, the most demanding cadence) into a
device-resident map of ~1 M live surfels.  Inputs are resident in HBM before the timed region.  Each
rank owns an independent sequence (weak scaling); the only inter-GPU traffic is one RCCL all_gather of
per-sequence counters after the timed loop.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md): 8.0 TB/s
W, H = 640, 480
SURFEL_BYTES = 56              # sizeof(Surfel), reference include/Surfel.h:28-37
K_FUSE = 7                     # kernel id of k_fuse in msl_sf_kernel_name()
ORB_READ_BYTES = 2 * 950532    # 2 * sum of the 8 pyramid level areas at 640x480, scale 1.2


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames-per-step", type=int, default=32)
    ap.add_argument("--surfels", type=int, default=1_000_000)
    ap.add_argument("--cpu-frames", type=int, default=128, help="frames of the bounded CPU-baseline sample, ~14 s on one core (0 = skip)")
    ap.add_argument("--no-breakdown", action="store_true")
    return ap.parse_args()


def build_inputs(rank, F, n_surfels):
    """F RGB-D frames of this rank's sequence + the pre-seeded live map (numpy, host)."""
    from manhattanslam_amd import synth, SURFEL_DTYPE
    grays, depths, poses = [], [], []
    member = None
    for f in range(F):
        _, depth, member, pose = synth.surfel_frame(f, seed=7 + 1000 * rank)
        grays.append(synth.orb_frame(synth.ORB_SEED + 1000 * rank + f))     # one gray image per frame, used by both stages
        depths.append(depth)
        poses.append(pose)
    smap = synth.surfel_map(n_surfels, ref=0, seed=11 + rank, min_update_times=5).astype(SURFEL_DTYPE)
    return np.stack(grays), np.stack(depths), member, poses, smap


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (separate FETCH_SIZE / WRITE_SIZE runs of this
    same command, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950): profiles/<tag>_summary.json."""
    import glob
    best = None
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_summary.json")), key=os.path.getmtime)
    cur = os.path.join(ROOT, "profiles", "current.txt")   # tag of the profile that matches the committed kernels
    if os.path.exists(cur):
        files.append(os.path.join(ROOT, "profiles", open(cur).read().strip() + "_summary.json"))
    for f in files:
        try:
            e = json.load(open(f)).get(kernel)
        except Exception:
            continue
        if e and "fetch_bytes_corrected" in e and "write_bytes" in e:
            best = (e["fetch_bytes_corrected"] + e["write_bytes"], os.path.basename(f))
    return best


def aggregate(local_ms, counters, world, device=None):
    """max-over-ranks of the timed region + all_gather of the per-sequence counters (RCCL on GPU, gloo on CPU)."""
    import torch
    import torch.distributed as dist
    if world == 1:
        return local_ms, [counters]
    t = torch.tensor([local_ms], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    c = torch.tensor(counters, dtype=torch.int64, device=device)
    out = [torch.zeros_like(c) for _ in range(world)]
    dist.all_gather(out, c)
    return float(t.item()), [o.tolist() for o in out]


def cpu_baseline(grays, depths, member, poses, smap, n_frames):
    """The CPU oracle (a port: the reference itself cannot be built without OpenCV/Eigen) on a bounded sample."""
    from tests import oracle_lib
    from manhattanslam_amd import synth
    o = oracle_lib.load()
    ex = o.orb_create(1000, 1.2, 8, 20, 7)
    I = synth.TUM1
    sf = oracle_lib.OracleSurfel(W, H, I["fx"], I["fy"], I["cx"], I["cy"], 30.0, 0.5)
    sf.map_set(smap)
    F = len(grays)
    t_orb = t_sf = 0.0
    for i in range(n_frames):
        f = i % F
        t0 = time.perf_counter()
        ex.extract(grays[f])
        t1 = time.perf_counter()
        sf.fuse_map(i, grays[f], depths[f], member, poses[f])
        t2 = time.perf_counter()
        t_orb += t1 - t0
        t_sf += t2 - t1
    tot = t_orb + t_sf
    return {"value": round(n_frames / tot, 3), "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": f"{n_frames} frames of the same workload (ORB + surfel fusion per frame, {len(smap)} seeded surfels), "
                      f"single thread, oracle/libmsl_oracle.so (g++ -O3, no -march=native)",
            "orb_ms_per_frame": round(1e3 * t_orb / n_frames, 2), "surfel_ms_per_keyframe": round(1e3 * t_sf / n_frames, 2),
            "host_cpus": os.cpu_count()}


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the front end has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)

    from manhattanslam_amd import ORBextractor, SurfelFusion, synth
    F = args.frames_per_step
    grays, depths, member, poses, smap = build_inputs(rank, F, args.surfels)
    I = synth.TUM1
    orb = ORBextractor(1000, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=F, device=local_rank)
    sf = SurfelFusion(W, H, I["fx"], I["fy"], I["cx"], I["cy"], 30.0, 0.5, device=local_rank)
    sf.set_batch_capacity(F)
    sf.map_reserve(2 * args.surfels + 65536)
    sf.map_upload(smap)

    d_gray = torch.from_numpy(grays).to(dev)
    d_depth = torch.from_numpy(depths).to(dev)
    d_member = torch.from_numpy(member).to(dev)
    cap = orb.capacity
    d_kps = torch.zeros(F * cap * 28, dtype=torch.uint8, device=dev)
    d_desc = torch.zeros(F * cap * 32, dtype=torch.uint8, device=dev)
    d_n = torch.zeros(F, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()

    frame_no = [0]

    def step():
        # ORB: one frame-batched launch sequence.  Surfel fusion: every frame is a keyframe; the superpixel stage of the
        # F keyframes is frame-batched, the map stage (fuse / new surfels / compaction) runs keyframe after keyframe.
        orb.extract_batch_device(d_gray, d_kps, d_desc, d_n, F, W, H)
        sf.fuse_resident_batch(np.arange(frame_no[0], frame_no[0] + F), d_gray, d_depth, d_member, poses, device=True,
                               member_shared=True)
        frame_no[0] += F

    def sync_all():
        orb.sync()
        sf.sync()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync_all()
    n_live_start = sf.counters()["n_live_after"]

    # ---- timed region: exactly K steps, bracketed by barrier + synchronize; only k_fuse carries HIP events ----
    sf.profile_enable(1 << K_FUSE)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync_all()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    local_ms = (time.perf_counter() - t0) * 1e3
    fuse_ms, fuse_launches = sf.profile_read()["k_fuse"]
    sf.profile_enable(0)
    ctr = sf.counters()
    n_kp = int(d_n.sum().item())
    counters = [args.steps * F, n_kp, ctr["n_live_after"], ctr["n_new"], ctr["n_updated"], ctr["n_deleted"], int(local_ms * 1e6), n_live_start]
    total_ms, gathered = aggregate(local_ms, counters, world, dev)

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    frames_total = args.steps * F * world
    value = frames_total / (total_ms * 1e-3)
    n_live_avg = 0.5 * (n_live_start + ctr["n_live_after"])
    fuse_s = fuse_ms * 1e-3 / max(fuse_launches, 1)
    achieved = SURFEL_BYTES * n_live_avg / fuse_s / 1e9 if fuse_launches else 0.0
    out = {
        "metric": "RGB-D frames/sec at 640x480 (ORB+surfel front end)",
        "value": round(value, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(total_ms / args.steps, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u8/i32 (ORB), f32+f64 (surfel)", "data": "synthetic",
        "config": {"workload": "ORB (1000 features, 8 levels, 1.2, FAST 20/7) + SurfelFusion on every frame (keyframe_every=1), "
                               "640x480, one independent sequence per GPU",
                   "frames_per_step": F, "seeded_surfels": args.surfels, "n_live_surfels": int(n_live_avg),
                   "intrinsics": "TUM1", "sequences_per_gpu": 1},
        "roofline": {"bound": "hbm", "kernel": "k_fuse", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": (pmc_traffic("k_fuse") or (None, None))[0],
                     "traffic_source": (pmc_traffic("k_fuse") or (None, "no committed PMC summary"))[1],
                     "algorithmic_bytes_per_launch": int(SURFEL_BYTES * n_live_avg), "avg_launch_us": round(fuse_s * 1e6, 2),
                     "timer": "HIP events carried by the k_fuse dispatch (hipExtLaunchKernelGGL) on the map stream, timed region",
                     "launches": int(fuse_launches)},
        # whole-pipeline algorithmic HBM reads per frame (ORB 2 * sum P_l + surfel 56 N + 5 W H + 4 (W/2)(H/2))
        # times the per-GPU frame rate, against the same peak
        "pipeline_roofline": {"algorithmic_read_bytes_per_frame": int(ORB_READ_BYTES + SURFEL_BYTES * n_live_avg + 5 * W * H + W * H),
                              "achieved": round((ORB_READ_BYTES + SURFEL_BYTES * n_live_avg + 6 * W * H) * value / world / 1e9, 1),
                              "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": round((ORB_READ_BYTES + SURFEL_BYTES * n_live_avg + 6 * W * H) * value / world / 1e9 / HBM_PEAK_GBS, 4)},
        "counters_per_rank": gathered,
    }

    if not args.no_breakdown:
        # per-kernel HIP-event breakdown (outside the timed region) + stage-only rates
        sf.profile_enable(-1)
        orb.profile_enable(-1)
        for _ in range(2):
            step()
        sync_all()
        nfr = 2 * F
        out["kernel_us_per_frame"] = {
            **{k: round(ms * 1e3 / nfr, 2) for k, (ms, c) in sf.profile_read().items() if c},
            **{"orb:" + k: round(ms * 1e3 / nfr, 2) for k, (ms, c) in orb.profile_read().items() if c}}
        sf.profile_enable(0)
        orb.profile_enable(0)
        t0 = time.perf_counter()
        for _ in range(10):
            orb.extract_batch_device(d_gray, d_kps, d_desc, d_n, F, W, H)
        orb.sync()
        out["orb_only_fps"] = round(10 * F / (time.perf_counter() - t0), 1)
        t0 = time.perf_counter()
        for _ in range(4):
            sf.fuse_resident_batch(np.arange(frame_no[0], frame_no[0] + F), d_gray, d_depth, d_member, poses, device=True, member_shared=True)
            frame_no[0] += F
        sf.sync()
        out["surfel_only_keyframes_per_sec"] = round(4 * F / (time.perf_counter() - t0), 1)

    if not args.no_breakdown:
        # the same kernel without co-running work: put the whole surfel pipeline on ONE stream (no overlap with the batched
        # superpixel stage or ORB) and time k_fuse again.  Reported next to, never instead of, the in-region roofline.
        sf.set_stream(torch.cuda.current_stream().cuda_stream)
        sf.profile_enable(1 << K_FUSE)
        for _ in range(2):
            sf.fuse_resident_batch(np.arange(frame_no[0], frame_no[0] + F), d_gray, d_depth, d_member, poses, device=True, member_shared=True)
            frame_no[0] += F
        ms_iso, n_iso = sf.profile_read()["k_fuse"]
        sf.profile_enable(0)
        n_now = sf.counters()["n_live_after"]
        iso = SURFEL_BYTES * n_now / (ms_iso * 1e-3 / max(n_iso, 1)) / 1e9
        out["roofline_isolated"] = {"kernel": "k_fuse", "achieved": round(iso, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                    "frac": round(iso / HBM_PEAK_GBS, 4), "avg_launch_us": round(ms_iso * 1e3 / max(n_iso, 1), 2),
                                    "note": "single stream, no co-running kernels; HIP events carried by the dispatch"}

    if args.cpu_frames > 0 and world == 1:
        out["cpu_baseline"] = cpu_baseline(grays, depths, member, poses, smap, args.cpu_frames)
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
