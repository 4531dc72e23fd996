#!/usr/bin/env python3
"""Benchmark of the MI355X-native RGB-D front end (ORB extractor + surfel fusion).

    python bench.py --gpus N --steps K --warmup W [--config frontend|2|3|4|5] [--io resident|host]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment re-launches itself under
torch.distributed.run (one rank per GPU, RCCL); the printed n_gpus always equals --gpus.

One *step* = P passes over one stationary batch of F synthetic RGB-D frames through the hot path on each GPU, inputs resident in
HBM.  A pass starts by putting the pre-seeded ~1 M-surfel map back (msl_sf_map_restore: a device-to-device copy INSIDE the timed
region, ~0.2 % of a pass) and restarts the keyframe numbering, so every pass -- and therefore every step -- does identical work:
N_live stays within ~1 % of --surfels and the surfels fused per keyframe stay constant whatever --steps is.  (The free-running
synthetic sequence of rounds 1-2 grew by ~40 surfels per keyframe and `value` moved 10 % with the run length; --no-reseed still
runs it.)

  --config frontend (default, the BASELINE.json metric): ORB extraction of the F gray frames (frame-batched launch sequences of
        --batch frames) + surfel fusion of every frame (keyframe_every = 1, the most demanding cadence) into a device-resident
        map of ~1 M live surfels, 640x480, TUM1 intrinsics.
  --config 2: ORBextractor only.                      --config 3: SurfelFusion only (~1 M live surfels).
  --config 4: ICL-NUIM intrinsics (fy = -480), ORB on every frame; on every k-th frame (--keyframe-every, default 4; the
        reference's cadence is data dependent, src/Tracking.cc:1433-1508) the PEAC plane extractor (msl_peac_membership_batch) and
        SurfelFusion with its membership image.
  --config 5: the frontend workload on 1280x960 frames (2x TUM1 intrinsics); meant for --gpus 8, one sequence per GPU.

  --io host: the same passes with every frame coming from pinned HOST memory and every ORB result going back to pinned host memory
        (keypoints, descriptors, counts), through the library's MSL_MEM_HOST paths; the line then carries `value_streaming` and the
        PCIe rates next to the resident `value` (SURVEY.md 7 hard part 7: every ORB consumer in src/Frame.cc:103-153 is host code).

Each rank owns an independent sequence (weak scaling); the only inter-GPU traffic is one RCCL all_gather of per-sequence
counters after the timed loop.  Rank 0 prints ONE JSON line.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

EVENT_STRIDE = 5      # k_fuse dispatches of the timed region that carry HIP events: one in five
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md): 8.0 TB/s
SURFEL_BYTES = 56              # sizeof(Surfel), reference include/Surfel.h:28-37

CONFIGS = {
    "frontend": dict(orb=True, sf=True, size="640x480", intr="TUM1", variant="A", kfe=1, passes=13, map="dense",
                     name="ORB (1000 features, 8 levels, 1.2, FAST 20/7) + SurfelFusion on every frame"),
    "2": dict(orb=True, sf=False, size="640x480", intr="TUM1", variant="A", kfe=1, passes=56, orb_batch=128,
              name="BASELINE config 2: ORBextractor only (128 frames per call: alone, its launch sequences amortise better than in 32-frame calls, 94.8 k vs "
                   "81.7 k frames/s; next to the surfel stage the call size makes no difference)"),
    "3": dict(orb=False, sf=True, size="640x480", intr="TUM1", variant="A", kfe=1, passes=15, map="dense", name="BASELINE config 3: SurfelFusion only"),
    "4": dict(orb=True, sf=True, size="640x480", intr="ICL", variant="A", kfe=4, peac=True, dropout=0.001, frames_per_pass=512, passes=7, map="dense", scene="clutter",
              name="BASELINE config 4: ICL-NUIM intrinsics (fy < 0), ORB every frame + PEAC plane extractor and SurfelFusion every k-th "
                   "frame (block fit and agglomerative clustering on the GPU, erosion / region growing on host threads; its membership "
                   "image feeds the fusion)"),
    "5": dict(orb=True, sf=True, size="1280x960", intr="TUM1", variant="A", kfe=1, passes=3, map="dense",
              name="BASELINE config 5: ORB + SurfelFusion on every frame, 1280x960 sequences"),
}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="frontend", choices=sorted(CONFIGS))
    ap.add_argument("--io", default="resident", choices=["resident", "host"], help="host: additionally time the passes with pinned host buffers in and out")
    ap.add_argument("--size", default=None, help="WxH override")
    ap.add_argument("--keyframe-every", type=int, default=None, help="SurfelFusion on every k-th frame")
    ap.add_argument("--frames-per-pass", type=int, default=0, help="frames of one pass = the stationary batch (each frame with its own input memory); the map "
                    "is put back at the start of every pass.  0 = the configuration's default (256; config 4: 512, because the plane extractor clusters "
                    "one keyframe per wave and is latency-bound per call, so its throughput grows with the keyframes handed over at once)")
    ap.add_argument("--passes-per-step", type=int, default=0, help="passes one step makes over the batch; 0 = the configuration's default, chosen so that "
                    "the default 20 steps time at least 3 s (frontend: 13 x 256 = 3328 frames per step)")
    ap.add_argument("--no-reseed", action="store_true", help="free-running sequence of rounds 1-2: never put the map back (not stationary)")
    ap.add_argument("--batch", type=int, default=32, help="frames per library call; a pass issues frames-per-pass / batch calls.  The scratch of "
                    "2 x batch keyframe slots plus the map should stay inside the 256 MB Infinity Cache: 128-frame batches measured 35 %% slower")
    ap.add_argument("--orb-batch", type=int, default=0, help="frames per ORB library call (0 = --batch); the ORB extractor has no per-frame dependency chain, "
                    "so its launch sequences amortise better over more frames than the surfel stage's 32-keyframe calls allow")
    ap.add_argument("--distinct-frames", type=int, default=64, help="distinct synthetic frames generated per sequence (SURVEY.md 8(d): 64), tiled to a pass")
    ap.add_argument("--surfels", type=int, default=1_000_000)
    ap.add_argument("--sequences-per-gpu", type=int, default=1, help="independent RGB-D sequences per GPU, each with its own handles, streams, resident map and host "
                    "thread (the weak-scaling unit stays the sequence; BASELINE's shape is 1).  `value` is the total over all sequences")
    ap.add_argument("--map", default=None, choices=["dense", "sparse", "moving"], help="pre-seeded live map: dense = ~35 %% of it inside the frustum of every keyframe (SURVEY.md "
                    "8(d) config 3 as written; the default of every configuration); sparse = the area-uniform room map of rounds 1-3 (~6 %% in view); "
                    "moving = a pan of 4 degrees per keyframe over a dense map with unmapped stripes: every keyframe spawns and deletes hundreds of "
                    "surfels (a pass is one sweep of --distinct-frames keyframes)")
    ap.add_argument("--map-order", default="creation", choices=["creation", "random"], help="array order of the dense map: creation = (source keyframe, superpixel) as "
                    "initializeSurfels appends surfels; random = no locality between array neighbours")
    ap.add_argument("--scene", default=None, choices=["room", "clutter"], help="depth / membership content: the bare box room or the furnished room (config 4's default)")
    ap.add_argument("--cpu-frames", type=int, default=48, help="frames of the single-thread CPU-baseline sample (0 = no CPU baseline)")
    ap.add_argument("--no-breakdown", action="store_true")
    ap.add_argument("--no-parity-gate", action="store_true", help="skip the parity gate behind the timed region (profiler passes: its few launches would enter the per-kernel "
                    "averages); the default line carries it and bench.py exits non-zero when it fails")
    ap.add_argument("--dry-run", action="store_true", help="launcher / aggregation / device-binding check on CPU (gloo), no GPU work, fabricated timings")
    return ap.parse_args(argv)


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(args):
    """`python bench.py --gpus N` (no rendezvous in the environment): become the launcher of N ranks."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def sequence_seeds(rank):
    from manhattanslam_amd import synth
    return synth.sequence_seeds(rank)


def build_inputs(rank, D, n_surfels, W, H, intr, cfg, args):
    """D distinct RGB-D frames of this rank's sequence + the pre-seeded live map (numpy, host): manhattanslam_amd.synth.bench_inputs."""
    from manhattanslam_amd import synth, SURFEL_DTYPE
    g, d, member, poses, smap = synth.bench_inputs(rank, D, n_surfels, W, H, intr, variant=cfg["variant"], dropout=cfg.get("dropout", 0.02),
                                                   map_kind=args.map or cfg.get("map", "dense"), map_order=args.map_order,
                                                   scene=args.scene or cfg.get("scene", "room"))
    return g, d, member, poses, smap.astype(SURFEL_DTYPE)


def kernel_source_hash():
    """Hash of the kernel sources: a committed PMC summary is quoted only for the kernels it was measured on."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "manhattanslam_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(kernel, config):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes of THIS command on THESE kernel sources (separate
    FETCH_SIZE / WRITE_SIZE runs, FETCH_SIZE corrected as MI355X_MICROARCH.md prescribes for gfx950): profiles/<tag>_summary.json.
    Returns (bytes, source) or (None, why)."""
    cur = os.path.join(ROOT, "profiles", "current.txt")   # tag of the profile that matches the committed kernels
    if not os.path.exists(cur):
        return None, "no profiles/current.txt"
    tag = open(cur).read().strip()
    f = os.path.join(ROOT, "profiles", f"{tag}_summary.json" if config == "frontend" else f"{tag}_config{config}_summary.json")
    try:
        doc = json.load(open(f))
    except Exception:  # noqa: BLE001
        return None, f"no committed PMC summary {os.path.basename(f)}"
    if doc.get("_meta", {}).get("kernel_source_hash") != kernel_source_hash():
        return None, f"{os.path.basename(f)} was measured on other kernel sources (hash mismatch): not quoted"
    e = doc.get(kernel)
    if e and "fetch_bytes_corrected" in e and "write_bytes" in e:
        return e["fetch_bytes_corrected"] + e["write_bytes"], os.path.basename(f)
    return None, f"{os.path.basename(f)} has no counters for {kernel}"


def rocprof_clock(kernel, config, alg_bytes):
    """The roofline fraction on rocprofv3's own clock: the kernel's average duration in the committed --kernel-trace --stats summary of THIS command
    on THESE kernel sources (profiles/<tag>_summary.json, hash-tied like pmc_traffic).  This is the figure to quote; the event-pair figure of the
    live run stands beside it.  None when no matching summary is committed."""
    cur = os.path.join(ROOT, "profiles", "current.txt")
    if not os.path.exists(cur):
        return None
    tag = open(cur).read().strip()
    f = os.path.join(ROOT, "profiles", f"{tag}_summary.json" if config == "frontend" else f"{tag}_config{config}_summary.json")
    try:
        doc = json.load(open(f))
    except Exception:  # noqa: BLE001
        return None
    e = doc.get(kernel)
    if doc.get("_meta", {}).get("kernel_source_hash") != kernel_source_hash() or not e or not e.get("avg_us"):
        return None
    return {"avg_launch_us": e["avg_us"], "calls": e.get("calls"), "frac": round(alg_bytes / (e["avg_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
            "source": os.path.basename(f), "note": "rocprofv3 --kernel-trace --stats of this command on these kernel sources (committed, hash-tied): the clock the roofline claim is made on"}


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


def cpu_baseline(args, cfg, W, H, kfe):
    """tools/cpu_baseline.py in its own interpreter: the oracle (a port: the reference cannot be built without OpenCV/Eigen)
    on bounded samples of the same workload -- 1 thread, the reference's 10-thread SurfelFusion, one sequence per host core."""
    cmd = [sys.executable, os.path.join(ROOT, "tools", "cpu_baseline.py"), "--frames", str(args.cpu_frames), "--size", f"{W}x{H}",
           "--intrinsics", cfg["intr"], "--surfels", str(args.surfels), "--keyframe-every", str(kfe), "--map", args.map or cfg.get("map", "dense"),
           "--map-order", args.map_order, "--scene", args.scene or cfg.get("scene", "room")]
    if not cfg["orb"]:
        cmd.append("--no-orb")
    if not cfg["sf"]:
        cmd.append("--no-surfel")
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
        res = json.loads(r.stdout.strip().splitlines()[-1])
    except Exception as e:  # noqa: BLE001
        return {"error": f"cpu baseline failed: {e!r}"}
    top = res.get("throughput") if "value" in res.get("throughput", {}) else res.get("single_thread", {})
    out = {"value": top.get("value"), "unit": top.get("unit", "frames/s"), "cores": top.get("cores"), "kind": "port", "sample": top.get("sample"),
           "host_cpus": res.get("host_cpus"), "usable_cpus": res.get("usable_cpus"), "code": res.get("code")}
    for k in ("single_thread", "surfel_10_threads", "throughput"):
        if k in res:
            out[k] = res[k]
    return out


GATE_FRAMES = 4       # ORB frames / keyframes the parity gate checks against the oracle after the timed region


def parity_gate(args, q, cfg, W, H, intr, kfe, B, nkf, local_rank):
    """SURVEY.md 8(d), last row ("parity gates in the same run"): after the timed region, GATE_FRAMES frames of this sequence's own workload go
    through the same handles once more -- ORB of its first frames; the pre-seeded map put back and its first keyframes fused (same call shape as
    the timed passes: device-resident inputs, batched call) -- and the results are checked against the CPU oracle in the cpu_baseline interpreter
    (tools/cpu_baseline.py --parity-gate; the only place bench.py meets oracle/).  Returns the checker's verdict; main() exits non-zero if it is not ok."""
    import tempfile
    import torch
    do_orb, do_sf = cfg["orb"], cfg["sf"]
    data = {"size": np.array([W, H], np.int32), "intr": np.array([intr["fx"], intr["fy"], intr["cx"], intr["cy"]], np.float64)}
    n = min(GATE_FRAMES, len(q.grays))
    if do_orb:
        cap = q.orb.capacity
        q.orb.extract_batch_device(q.d_gray, q.d_kps, q.d_desc, q.d_n, n, W, H)
        q.orb.sync()
        data["orb_gray"] = q.grays[:n]
        data["orb_n"] = q.d_n[:n].cpu().numpy()
        data["orb_kps"] = q.d_kps[:n * cap * 28].cpu().numpy().reshape(n, cap * 28)
        data["orb_desc"] = q.d_desc[:n * cap * 32].cpu().numpy().reshape(n, cap * 32)
    if do_sf:
        nk = min(GATE_FRAMES, nkf)
        q.sf.sync()
        q.sf.map_restore()
        refs = np.arange(nk)
        use_peac = bool(cfg.get("peac"))
        if use_peac:
            if q.peac_dev[0] is None:
                q.sub_sf(0); q.sf.sync(); q.sf.map_restore()          # (a dry call that fetches the plane membership of this pass's keyframes)
            member_d, member_h, shared = q.peac_dev[0][:nk], q.peac_dev[0][:nk].cpu().numpy(), False
        else:
            member_d, member_h, shared = q.d_member, q.member[None], True
        q.sf.fuse_resident_batch(refs, q.d_gray, q.d_depth, member_d, q.kf_poses[0][:nk], device=True, member_shared=shared, frame_step=kfe,
                                 **({"member_frame_step": 1} if use_peac else {}))
        q.sf.sync()
        D = len(q.grays)
        fr = [(j * kfe) % D for j in range(nk)]
        data.update(sf_map0=q.smap.view(np.uint8).reshape(len(q.smap), -1), sf_map_gpu=q.sf.map_download().view(np.uint8).reshape(-1, q.smap.dtype.itemsize),
                    sf_refs=refs, sf_gray=q.grays[fr], sf_depth=q.depths[fr], sf_member=member_h, sf_poses=np.stack([np.asarray(p, np.float32).reshape(16) for p in q.kf_poses[0][:nk]]))
        q.sf.map_restore()
        q.kf_no[0] = 0
    fd, path = tempfile.mkstemp(prefix=f"msl_gate_{local_rank}_", suffix=".npz", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    os.close(fd)
    try:
        np.savez(path, **data)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "cpu_baseline.py"), "--parity-gate", path], capture_output=True, text=True, timeout=600)
        try:
            res = json.loads(r.stdout.strip().splitlines()[-1])
        except Exception:  # noqa: BLE001
            res = {"ok": False, "failures": [f"parity checker failed to run: rc {r.returncode}, {r.stderr[-400:]!r}"]}
    finally:
        if os.path.exists(path):
            os.unlink(path)
    return res


def geometry(args):
    cfg = CONFIGS[args.config]
    W, H = (int(v) for v in (args.size or cfg["size"]).lower().split("x"))
    kfe = args.keyframe_every or cfg["kfe"]
    F = args.frames_per_pass or cfg.get("frames_per_pass", 256)
    P = args.passes_per_step or cfg["passes"]
    if (args.map or cfg.get("map")) == "moving" and not args.frames_per_pass:
        # the moving-camera regime: a pass is ONE sweep over the distinct frames (a 4 degree pan per keyframe into partly unmapped territory), never
        # a repetition of it -- a frame seen again would find the surfels it spawned the first time
        F = min(args.distinct_frames, F)
        if not args.passes_per_step:
            P = P * max(1, cfg.get("frames_per_pass", 256) // F)
    B = min(args.batch, F)
    D = min(args.distinct_frames, F)
    if F % B or B % kfe or F % D:
        raise SystemExit("--frames-per-pass must be a multiple of --batch and --distinct-frames, --batch a multiple of --keyframe-every")
    return cfg, W, H, kfe, F, P, B, D


def dry_run(args, world, rank, local_rank):
    """Launcher / aggregation / binding path without a GPU: gloo, fabricated per-rank timings; every rank reports the device it
    would bind (LOCAL_RANK -> hipSetDevice(local_rank) in both handles, independent of HIP_VISIBLE_DEVICES) and its sequence seeds."""
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group("gloo")
        dist.barrier()
    cfg, W, H, kfe, F, P, B, D = geometry(args)
    local_ms = 10.0 + 5.0 * rank
    S = max(1, args.sequences_per_gpu)
    sd = sequence_seeds(rank * S)                      # this rank's sequences are the global sequences rank * S .. rank * S + S - 1
    counters = [args.steps * P * F * S, 1000 * (rank + 1), 990000 + rank, 3, 50000, 7, int(local_ms * 1e6), 1000000]
    binding = [rank, local_rank, local_rank, sd["frame"], sd["orb"], sd["map"], int(os.environ.get("LOCAL_WORLD_SIZE", world)), S]
    total_ms, gathered = aggregate(local_ms, counters, world, None)
    _, bindings = aggregate(local_ms, binding, world, None)
    if rank == 0:
        frames = sum(g[0] for g in gathered)
        print(json.dumps({"metric": f"RGB-D frames/sec at {W}x{H} (ORB+surfel front end)", "value": round(frames / (total_ms * 1e-3), 1),
                          "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(total_ms / args.steps, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                          "dry_run": True, "data": "none (fabricated timings: launcher / gloo aggregation / binding check only)",
                          "config": {"config": args.config, "frames_per_pass": F, "passes_per_step": P},
                          "counters_per_rank": gathered,
                          "binding_per_rank": [dict(zip(("rank", "local_rank", "device", "frame_seed", "orb_seed", "map_seed", "local_world_size", "sequences"), b)) for b in bindings],
                          "sequences_per_gpu": S, "global_sequences_per_rank": [[b[0] * S + i for i in range(S)] for b in bindings]}),
              flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: refusing to report a line whose n_gpus differs from --gpus")
    if args.dry_run:
        return dry_run(args, world, rank, local_rank)
    if args.sequences_per_gpu > 1 or args.io == "host":
        # every sequence brings six streams of three priorities (the streaming passes: a second ORB handle and three host threads); the runtime maps
        # them onto GPU_MAX_HW_QUEUES (default 4) hardware queues, and streams that share a queue serialise: two sequences 15.8 k frames/s with 4
        # queues, 19.9 k with 8 (16: the same); --io host 16.1 k -> 17.6 k.  One resident sequence -- `value` -- is the same with 4, 8 or 16
        # (23.32 / 23.26 / 23.32 k).  Read when the runtime starts.
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the front end has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)

    from manhattanslam_amd import ORBextractor, SurfelFusion, synth
    cfg, W, H, kfe, F, P, B, D = geometry(args)
    do_orb, do_sf = cfg["orb"], cfg["sf"]
    reseed = do_sf and not args.no_reseed
    nsub = F // B
    nkf = B // kfe if do_sf else 0   # keyframes per library call
    OB = args.orb_batch or cfg.get("orb_batch", B)          # frames per ORB call
    if OB % B or F % OB:
        raise SystemExit("--orb-batch must be a multiple of --batch and divide --frames-per-pass")
    intr = synth.scaled_intrinsics(getattr(synth, cfg["intr"]), W)
    S = max(1, args.sequences_per_gpu)
    map_kind, scene_kind = args.map or cfg.get("map", "dense"), args.scene or cfg.get("scene", "room")
    use_peac = bool(cfg.get("peac")) and do_sf

    def make_seq(s_idx):
        """One independent RGB-D sequence on this GPU: its own inputs (seeds of global sequence rank * S + s_idx), extractor and fusion handles
        (own streams, own resident map) and the closures that enqueue its passes."""
        grays, depths, member, poses, smap = build_inputs(rank * S + s_idx, D, args.surfels if do_sf else 16, W, H, intr, cfg, args)

        orb = sf = None
        d_gray = torch.from_numpy(grays).to(dev).repeat(F // D, 1, 1).contiguous()
        if do_orb:
            orb = ORBextractor(1000, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=OB, device=local_rank)
            cap = orb.capacity
            d_kps = torch.zeros(F * cap * 28, dtype=torch.uint8, device=dev)
            d_desc = torch.zeros(F * cap * 32, dtype=torch.uint8, device=dev)
            d_n = torch.zeros(F, dtype=torch.int32, device=dev)
        if do_sf:
            sf = SurfelFusion(W, H, intr["fx"], intr["fy"], intr["cx"], intr["cy"], 30.0, 0.5, device=local_rank)
            sf.set_batch_capacity(nkf)
            sf.map_reserve(2 * args.surfels + 65536)
            sf.map_upload(smap)
            sf.map_snapshot()
            d_depth = torch.from_numpy(depths).to(dev).repeat(F // D, 1, 1).contiguous()
            d_member = torch.from_numpy(member).to(dev)
            if use_peac:
                from manhattanslam_amd import peac
                # raw 16-bit depth of the keyframes (5000 units per metre), resident in HBM like the other inputs
                d_depth16 = torch.from_numpy(np.stack([synth.depth_u16(d) for d in depths]).view(np.int16)).to(dev).repeat(F // D, 1, 1).contiguous()
                peac_prm = peac.default_params()
                # The extractor of the NEXT pass's keyframes runs on a host thread (block fit on the GPU, clustering on the library's worker
                # threads) while this pass's ORB / SurfelFusion calls are enqueued, the way the reference runs plane extraction in the tracking
                # thread and surfel mapping in its own (src/Tracking.cc:228, src/SurfelMapping.cpp:46-60).  Every timed pass still pays one full
                # extraction of its nkf * nsub keyframes: the one in flight at the end of the region is waited for inside it (sync_all).
                import concurrent.futures
                h_member = [np.zeros((nkf * nsub, H // 2, W // 2), np.int32) for _ in range(2)]
                h_nplanes = [np.zeros(nkf * nsub, np.int32) for _ in range(2)]
                peac_pool = concurrent.futures.ThreadPoolExecutor(1)
                peac_job = [None, 0]   # (future, buffer index)

                def peac_submit():
                    b = peac_job[1] ^ 1
                    peac_job[0] = peac_pool.submit(peac.plane_membership_device, d_depth16, kfe, nkf * nsub, W, H, intr["fx"], intr["fy"], intr["cx"], intr["cy"],
                                                   np.float32(1.0 / 5000.0), peac_prm, h_member[b], h_nplanes[b], local_rank)
                    peac_job[1] = b
            kf_poses = [[poses[(sb * B + j * kfe) % D] for j in range(nkf)] for sb in range(nsub)]
        torch.cuda.synchronize()

        kf_no = [0]
        peac_dev = [None]
        cap_ = orb.capacity if do_orb else 0

        def sub_orb(sb):   # the ORB call that starts at surfel call sb (one ORB call covers OB / B surfel calls)
            if (sb * B) % OB == 0:
                orb.extract_batch_device(d_gray[sb * B:], d_kps[sb * B * cap_ * 28:], d_desc[sb * B * cap_ * 32:], d_n[sb * B:], OB, W, H)

        def sub_sf(sb):
            if use_peac:
                if sb == 0:
                    # plane membership of this pass's keyframes (one call: block fit on the GPU, clustering on one host thread per keyframe),
                    # computed while the previous pass was enqueued; back to HBM, then start the next pass's
                    if peac_job[0] is None:
                        peac_submit()
                    peac_job[0].result()
                    peac_dev[0] = torch.from_numpy(h_member[peac_job[1]]).to(dev)
                    peac_submit()
                sf.fuse_resident_batch(np.arange(kf_no[0], kf_no[0] + nkf), d_gray[sb * B:], d_depth[sb * B:], peac_dev[0][sb * nkf:], kf_poses[sb], device=True,
                                       member_shared=False, frame_step=kfe, member_frame_step=1)
                kf_no[0] += nkf
                return
            # the superpixel stage of a call's keyframes is frame-batched, the map stage runs keyframe after keyframe
            sf.fuse_resident_batch(np.arange(kf_no[0], kf_no[0] + nkf), d_gray[sb * B:], d_depth[sb * B:], d_member, kf_poses[sb], device=True,
                                   member_shared=True, frame_step=kfe)
            kf_no[0] += nkf

        def begin_pass():
            if reseed:           # the pre-seeded map again (device-to-device, asynchronous on the map stream), keyframe numbering from 0
                sf.map_restore()
                kf_no[0] = 0

        def pass_orb():
            for sb in range(nsub):
                sub_orb(sb)

        def pass_sf():
            begin_pass()
            for sb in range(nsub):
                sub_sf(sb)

        def one_pass():
            begin_pass()
            for sb in range(nsub):
                if do_orb:
                    sub_orb(sb)
                if do_sf:
                    sub_sf(sb)

        def step():
            for _ in range(P):
                one_pass()

        def sync_all():
            if use_peac and peac_job[0] is not None:
                peac_job[0].result()
            if do_orb:
                orb.sync()
            if do_sf:
                sf.sync()
            torch.cuda.synchronize()

        import types
        return types.SimpleNamespace(**{k: v for k, v in locals().items() if k != "types"})

    seqs = [make_seq(i) for i in range(S)]
    q0 = seqs[0]    # the primary sequence: carries the roofline kernel's events and the per-kernel breakdown
    grays, depths, member, poses, smap, orb, sf = q0.grays, q0.depths, q0.member, q0.poses, q0.smap, q0.orb, q0.sf
    step, one_pass, pass_orb, pass_sf, begin_pass, sub_sf = q0.step, q0.one_pass, q0.pass_orb, q0.pass_sf, q0.begin_pass, q0.sub_sf
    d_n = q0.d_n if do_orb else None
    nkf_nsub = nkf * nsub

    def run_steps(k):
        """k steps of every sequence: one host thread per sequence beyond the first (the library calls release the GIL)."""
        if S == 1:
            for _ in range(k):
                step()
            return
        import threading
        th = [threading.Thread(target=lambda q=q: [q.step() for _ in range(k)]) for q in seqs]
        for t in th:
            t.start()
        for t in th:
            t.join()

    def sync_all():
        for q in seqs:
            q.sync_all()

    run_steps(args.warmup)
    sync_all()
    n_live_start = int(len(smap)) if reseed else (sf.counters()["n_live_after"] if do_sf else 0)
    tot0 = sf.debug_ctr() if do_sf else None

    # ---- timed region: exactly K steps, bracketed by barrier + synchronize; only the roofline kernel carries HIP events ----
    sf_names = {v: k for k, v in enumerate(sf.kernel_names())} if do_sf else {}
    orb_names = {v: k for k, v in enumerate(orb.kernel_names())} if do_orb else {}
    roof_kernel = "k_fuse" if do_sf else "k_fast"
    if do_sf:
        # every EVENT_STRIDE-th k_fuse dispatch carries events (a stride coprime with the keyframes per call: every place of the chain is sampled
        # equally); an event-carrying dispatch costs the stream ~0.3 us, and with ALL 66 560 launches timed `value` read 2-4 % lower
        sf.profile_stride(EVENT_STRIDE)
        sf.profile_enable((1 << sf_names[roof_kernel]) | (1 << sf_names["k_empty"]))   # + empty dispatches at the same place of the chain: the event pair's own time
    else:
        orb.profile_enable(1 << orb_names[roof_kernel])
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_steps(args.steps)
    sync_all()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    local_ms = (time.perf_counter() - t0) * 1e3
    empty_us = None
    if do_sf:
        prof = sf.profile_read()
        roof_ms, roof_launches = prof[roof_kernel]
        if prof["k_empty"][1]:
            empty_us = prof["k_empty"][0] * 1e3 / prof["k_empty"][1]
        sf.profile_enable(0)
        sf.profile_stride(1)
    else:
        roof_ms, roof_launches = orb.profile_read()[roof_kernel]
        orb.profile_enable(0)
    ctr = sf.counters() if do_sf else dict(n_live_after=0, n_new=0, n_updated=0, n_deleted=0)
    tot1 = sf.debug_ctr() if do_sf else None
    n_kp_pass = sum(int(q.d_n.sum().item()) for q in seqs) if do_orb else 0       # all sequences of this GPU
    frames_rank = args.steps * P * F * S
    counters = [frames_rank, n_kp_pass * args.steps * P, ctr["n_live_after"], ctr["n_new"], ctr["n_updated"], ctr["n_deleted"], int(local_ms * 1e6), n_live_start]
    total_ms, gathered = aggregate(local_ms, counters, world, dev)
    # parity gate (outside the timed region): every rank checks its own sequence; the verdicts travel like the counters
    gate = None
    if not args.no_parity_gate:
        gate = parity_gate(args, q0, cfg, W, H, intr, kfe, B, nkf, local_rank)
        _, gate_all = aggregate(0.0, [1 if gate.get("ok") else 0, int(gate.get("orb_frames", 0)), int(gate.get("keyframes", 0)), int(1e12 * float(gate.get("max_abs", 0.0)))], world, dev)
        gate["ok_all_ranks"] = all(g[0] == 1 for g in gate_all)
        gate["per_rank_ok"] = [bool(g[0]) for g in gate_all]
        gate["max_abs_all_ranks"] = max(g[3] for g in gate_all) * 1e-12

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        if gate is not None and not gate.get("ok"):
            raise SystemExit(3)
        return

    frames_total = frames_rank * world
    value = frames_total / (total_ms * 1e-3)
    # per-keyframe averages over the timed region from the handle's running totals (new, deleted, updated, keyframes, live before)
    if do_sf:
        dk = max(int(tot1[11] - tot0[11]), 1)
        avg_new, avg_del, avg_upd, n_live_avg = (float(tot1[i] - tot0[i]) / dk for i in (8, 9, 10, 12))
    else:
        dk, avg_new, avg_del, avg_upd, n_live_avg = 0, 0.0, 0.0, 0.0, 0.0
    sum_pl = sum(w * h for w, h in (orb.level_size(l) for l in range(8))) if do_orb else 0   # sum of the pyramid level areas
    # An event pair carried by a dispatch (hipExtLaunchKernelGGL) spans from the completion of the PREVIOUS command on the stream to the end of
    # the kernel: it contains the dependent-launch gap.  An EMPTY kernel timed the same way at the same place of the chain gives that part
    # (~4 us); the difference is the kernel's execution time, which is what rocprofv3 --kernel-trace reports (profiles/<tag>_summary.md).
    raw_launch_s = roof_ms * 1e-3 / max(roof_launches, 1)
    launch_s = max(raw_launch_s - (empty_us or 0.0) * 1e-6, 1e-9)
    # algorithmic bytes per launch of the roofline kernel: k_fuse reads the live surfels (56 B each) once per
    # keyframe; k_fast reads every pyramid level of the OB frames of a call once
    alg_bytes = SURFEL_BYTES * n_live_avg if do_sf else float(sum_pl * OB)
    alg_write = SURFEL_BYTES * avg_upd if do_sf else 0.0
    achieved = alg_bytes / launch_s / 1e9 if roof_launches else 0.0
    achieved_rw = (alg_bytes + alg_write) / launch_s / 1e9 if roof_launches else 0.0
    # R = 2 sum P_l per frame + (56 N_live + 5 W H + 4 (W/2)(H/2)) per keyframe; W = (sum_{l>=1} P_l + sum P_l + 60 N_kp)
    # per frame + 56 (N_updated + N_new) per keyframe
    n_kp_frame = n_kp_pass / (F * S) if do_orb else 0.0
    r_frame = (2 * sum_pl if do_orb else 0) + ((SURFEL_BYTES * n_live_avg + 5 * W * H + W * H) / kfe if do_sf else 0)
    w_frame = ((2 * sum_pl - W * H + 60 * n_kp_frame) if do_orb else 0) + (SURFEL_BYTES * (avg_upd + avg_new) / kfe if do_sf else 0)
    traffic, traffic_src = pmc_traffic(roof_kernel, args.config)
    fps_gpu = value / world
    out = {
        "metric": f"RGB-D frames/sec at {W}x{H} (ORB+surfel front end)" if do_orb and do_sf else
                  (f"frames/sec at {W}x{H} (ORBextractor only)" if do_orb else f"keyframes/sec at {W}x{H} (SurfelFusion only)"),
        "value": round(value, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(total_ms / args.steps, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "/".join((["u8/i32 (ORB)"] if do_orb else []) + (["f32+f64 (surfel)"] if do_sf else [])), "data": "synthetic",
        "config": {"workload": f"{cfg['name']} (keyframe_every={kfe}), {W}x{H}, one independent sequence per GPU",
                   "config": args.config, "frames_per_step": F * P, "frames_per_pass": F, "passes_per_step": P, "frames_per_call": B, "orb_frames_per_call": OB if do_orb else 0,
                   "keyframes_per_pass": nkf * nsub, "keyframe_every": kfe, "distinct_frames": D,
                   "stationary": bool(reseed) or not do_sf,
                   "map_reseed": "msl_sf_map_restore (device-to-device, inside the timed region) at the start of every pass" if reseed else "none",
                   "map_stage": ("deferred compaction (MSL_SF_DEFER=1: one k_fuse launch per keyframe, windows of 32 replayed)" if os.environ.get("MSL_SF_DEFER") == "1"
                                 else "classic chain (k_fuse + k_compact per keyframe: the library's choice for a handle with its own two streams)") if do_sf else None,
                   "seeded_surfels": args.surfels if do_sf else 0, "n_live_surfels": int(n_live_avg),
                   "map": (f"{map_kind} ({'~35 % of the map inside the frustum of every keyframe, SURVEY.md 8(d) config 3' if map_kind == 'dense' else 'a 4 degree pan per keyframe over a dense map with unmapped stripes (synth.surfel_map_moving): 25-34 % in view, every keyframe spawns and deletes hundreds of surfels' if map_kind == 'moving' else 'area-uniform over the room, ~6 % in view'}"
                           f"{', array order = ' + args.map_order if map_kind == 'dense' else ''})") if do_sf else None,
                   "in_view_fraction_keyframe0": round(synth.in_view_fraction(smap, 0, W, H, intr), 4) if do_sf else None, "scene": scene_kind,
                   "surfels_updated_per_keyframe": round(avg_upd, 1), "surfels_new_per_keyframe": round(avg_new, 2), "surfels_deleted_per_keyframe": round(avg_del, 2),
                   "intrinsics": cfg["intr"], "membership": "PEAC plane extractor (msl_peac_membership_batch)" if use_peac else cfg["variant"],
                   "sequences_per_gpu": S, "per_sequence_ms_per_pass": round(total_ms / (args.steps * P), 4), "timed_region_s": round(total_ms * 1e-3, 4), "timed_frames_per_gpu": frames_rank},
        "roofline": {"bound": "hbm", "kernel": roof_kernel, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                     "traffic_source": traffic_src,
                     "traffic_gbs": round(traffic / launch_s / 1e9, 1) if traffic and roof_launches else None,
                     "algorithmic_bytes_per_launch": int(alg_bytes), "avg_launch_us": round(launch_s * 1e6, 2),
                     "avg_launch_us_event_pair_raw": round(raw_launch_s * 1e6, 2), "event_pair_empty_kernel_us": round(empty_us, 2) if empty_us else None,
                     "frac_on_raw_event_time": round(alg_bytes / raw_launch_s / 1e9 / HBM_PEAK_GBS, 4) if roof_launches else 0.0,
                     "rocprofv3": rocprof_clock(roof_kernel, args.config, alg_bytes),
                     "read_write": {"algorithmic_bytes_per_launch": int(alg_bytes + alg_write), "achieved": round(achieved_rw, 1),
                                    "frac": round(achieved_rw / HBM_PEAK_GBS, 4)},
                     "timer": f"HIP events carried by the {roof_kernel} dispatch (hipExtLaunchKernelGGL) on its own stream, "
                              + (f"every {EVENT_STRIDE}th launch of the timed region (rotating over the keyframes of a call), " if do_sf else "every launch of the timed region, ") +
                              "minus the time the same kind of event pair reports for an empty kernel launched at the same place of the chain "
                              "(same region): the pair's first event completes with the previous command, so the raw time contains the "
                              "dependent-launch gap; the difference is the kernel's execution time as rocprofv3 --kernel-trace reports it",
                     "launches": int(roof_launches)},   # launches that carried events
        # whole-pipeline algorithmic HBM bytes per frame times the per-GPU frame rate, against the same peak
        "pipeline_roofline": {"algorithmic_read_bytes_per_frame": int(r_frame), "algorithmic_write_bytes_per_frame": int(w_frame),
                              "achieved": round(r_frame * fps_gpu / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": round(r_frame * fps_gpu / 1e9 / HBM_PEAK_GBS, 4),
                              "read_write": {"achieved": round((r_frame + w_frame) * fps_gpu / 1e9, 1),
                                             "frac": round((r_frame + w_frame) * fps_gpu / 1e9 / HBM_PEAK_GBS, 4)}},
        "counters_per_rank": gathered,
        "parity_gate": gate,
    }

    if args.io == "host" and world == 1:
        out["streaming"] = streaming(args, cfg, grays, depths, member, poses, smap, W, H, intr, kfe, F, P, B, D, local_rank, value)

    if not args.no_breakdown:
        # per-kernel HIP-event breakdown (outside the timed region) + stage-only rates
        if do_sf:
            sf.profile_enable(-1)
        if do_orb:
            orb.profile_enable(-1)
        one_pass()
        sync_all()
        out["kernel_us_per_frame"] = {
            **({k: round(ms * 1e3 / F, 2) for k, (ms, c) in sf.profile_read().items() if c} if do_sf else {}),
            **({"orb:" + k: round(ms * 1e3 / F, 2) for k, (ms, c) in orb.profile_read().items() if c} if do_orb else {})}
        if do_sf:
            sf.profile_enable(0)
        if do_orb:
            orb.profile_enable(0)
            t0 = time.perf_counter()
            for _ in range(4):
                pass_orb()
            orb.sync()
            out["orb_only_fps"] = round(4 * F / (time.perf_counter() - t0), 1)
        if do_sf:
            # median of five single passes (each enqueued, then drained): right after the ORB-only passes above ONE pass of this loop now and then
            # takes 50-60 ms instead of 9.5 (the device, not the enqueue: a queue-scheduling hiccup at the change of the set of busy streams;
            # the timed region above shows nothing of the kind), which a two-pass mean turned into 7-27 k keyframes/s from run to run
            ts = []
            for _ in range(5):
                t0 = time.perf_counter()
                pass_sf()
                sf.sync()
                ts.append(time.perf_counter() - t0)
            out["surfel_only_keyframes_per_sec"] = round(nkf * nsub / sorted(ts)[2], 1)
            out["surfel_only_pass_ms"] = [round(1e3 * t, 2) for t in ts]
            # the same kernel without co-running work: the whole surfel pipeline on ONE stream (no overlap with the batched
            # superpixel stage or ORB), k_fuse timed again.  Reported next to, never instead of, the in-region roofline.
            sf.set_stream(torch.cuda.current_stream().cuda_stream)
            begin_pass()
            t_a = sf.debug_ctr()
            sf.profile_enable((1 << sf_names["k_fuse"]) | (1 << sf_names["k_empty"]))
            sub_sf(0)
            pi = sf.profile_read()
            ms_iso, n_iso = pi["k_fuse"]
            ms_iso -= n_iso * (pi["k_empty"][0] / max(pi["k_empty"][1], 1))      # the event pair's own time (see roofline.timer)
            sf.profile_enable(0)
            t_b = sf.debug_ctr()
            n_iso_live = float(t_b[12] - t_a[12]) / max(int(t_b[11] - t_a[11]), 1)
            iso = SURFEL_BYTES * n_iso_live / (ms_iso * 1e-3 / max(n_iso, 1)) / 1e9
            out["roofline_isolated"] = {"kernel": "k_fuse", "achieved": round(iso, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                        "frac": round(iso / HBM_PEAK_GBS, 4), "avg_launch_us": round(ms_iso * 1e3 / max(n_iso, 1), 2),
                                        "note": "single stream, no co-running kernels; HIP events carried by the dispatch"}
        if world == 1:
            out["dropin"] = dropin_shapes(grays, depths, member, poses, smap, W, H, intr, do_orb, do_sf, local_rank)

    if args.cpu_frames > 0 and world == 1:
        out["cpu_baseline"] = cpu_baseline(args, cfg, W, H, kfe)
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()
    if gate is not None and not gate.get("ok_all_ranks", gate.get("ok")):
        raise SystemExit(3)   # results differ from the oracle's: the line above says where (parity_gate.failures); no headline without parity


def streaming(args, cfg, grays, depths, member, poses, smap, W, H, intr, kfe, F, P, B, D, device, resident_value):
    """--io host: the same passes, but every call reads its frames from PINNED HOST memory and ORB writes keypoints / descriptors / counts
    back to pinned host memory, as the reference's consumers need them (src/Frame.cc:103-153 are host code), through the library's
    MSL_MEM_HOST paths.  Overlap of copies and kernels across calls comes from handles and threads, not from a new entry point: two ORB
    handles (own stream and staging each) are fed by two host threads with alternating calls -- a synchronous host-to-host call blocks
    only its thread --, and one thread feeds the surfel handle, whose host-image batches are asynchronous already (double-buffered slot
    sets, copies on the superpixel stream).  Reported next to `value`, never instead of it."""
    import threading
    import torch
    from manhattanslam_amd import ORBextractor, SurfelFusion
    do_orb, do_sf = cfg["orb"], cfg["sf"]
    nsub = F // B
    nkf = B // kfe if do_sf else 0
    reps = F // D
    g_np = torch.from_numpy(np.tile(grays, (reps, 1, 1))).pin_memory().numpy()     # torch pinned memory = hipHostMalloc
    orbs, sf = [], None
    if do_orb:
        orbs = [ORBextractor(1000, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=B, device=device) for _ in range(2)]
        cap = orbs[0].capacity
        h_kps = torch.zeros(F * cap * 28, dtype=torch.uint8).pin_memory().numpy()
        h_desc = torch.zeros(F * cap * 32, dtype=torch.uint8).pin_memory().numpy()
        h_n = torch.zeros(F, dtype=torch.int32).pin_memory().numpy()
    if do_sf:
        sf = SurfelFusion(W, H, intr["fx"], intr["fy"], intr["cx"], intr["cy"], 30.0, 0.5, device=device)
        sf.set_batch_capacity(nkf)
        sf.map_reserve(2 * args.surfels + 65536)
        sf.map_upload(smap)
        sf.map_snapshot()
        d_np = torch.from_numpy(np.tile(depths, (reps, 1, 1))).pin_memory().numpy()
        # the same depth as the sensor / the data set delivers it: raw 16-bit values, 5000 per metre (TUM / ICL), converted on the device
        d16_np = torch.from_numpy(np.tile(np.clip(np.rint(depths * 5000.0), 0, 65535).astype(np.uint16), (reps, 1, 1))).pin_memory().numpy()
        d16_factor = float(np.float32(1.0) / np.float32(5000.0))
        m_np = torch.from_numpy(member).pin_memory().numpy()
        kf_poses = [[poses[(sb * B + j * kfe) % D] for j in range(nkf)] for sb in range(nsub)]

    def orb_worker(which, npass):
        ex = orbs[which]
        for _ in range(npass):
            for sb in range(which, nsub, 2):
                ex.extract_batch_host(g_np[sb * B:(sb + 1) * B], h_kps[sb * B * cap * 28:], h_desc[sb * B * cap * 32:], h_n[sb * B:], B, W, H)

    def sf_worker(npass, raw16):
        for _ in range(npass):
            sf.map_restore()
            for sb in range(nsub):
                sf.fuse_resident_batch(np.arange(sb * nkf, (sb + 1) * nkf), g_np[sb * B:], (d16_np if raw16 else d_np)[sb * B:], m_np, kf_poses[sb],
                                       device=False, member_shared=True, frame_step=kfe, depth_factor=d16_factor if raw16 else None)
        sf.sync()

    def run(npass, raw16=False):
        th = [threading.Thread(target=orb_worker, args=(i, npass)) for i in range(len(orbs))]
        if do_sf:
            th.append(threading.Thread(target=sf_worker, args=(npass, raw16)))
        for t in th:
            t.start()
        for t in th:
            t.join()

    run(2)
    npass = max(2, min(P * args.steps, 24))
    t0 = time.perf_counter()
    run(npass)
    dt = time.perf_counter() - t0
    fps = npass * F / dt
    n_kp = int(h_n.sum()) if do_orb else 0
    h2d = (W * H if do_orb else 0) + ((W * H + 4 * W * H) / kfe if do_sf else 0)        # bytes per frame (gray for ORB; gray + f32 depth per keyframe)
    d2h = (60.0 * cap + 4) if do_orb else 0.0                                            # the library copies the full-capacity keypoint / descriptor rows
    res = {"value_streaming": round(fps, 1), "unit": "frames/s", "fraction_of_resident": round(fps / resident_value, 3), "passes": npass,
           "keypoints_per_frame": round(n_kp / F, 1) if do_orb else 0,
           "h2d_bytes_per_frame": int(h2d), "d2h_bytes_per_frame": int(d2h), "h2d_gbs": round(h2d * fps / 1e9, 2), "d2h_gbs": round(d2h * fps / 1e9, 3),
           "note": "pinned host buffers in (gray for ORB and again for SurfelFusion, f32 depth; the shared membership image once per call) and out "
                   "(keypoints, descriptors, counts); msl_orb_extract_batch / msl_sf_fuse_resident_batch with MSL_MEM_HOST; 2 ORB handles on 2 host "
                   "threads + 1 thread for the surfel handle"}
    if do_sf:
        run(2, True)
        t0 = time.perf_counter()
        run(npass, True)
        fps16 = npass * F / (time.perf_counter() - t0)
        h2d16 = (W * H if do_orb else 0) + (W * H + 2 * W * H) / kfe
        res["raw_depth16"] = {"value_streaming": round(fps16, 1), "fraction_of_resident": round(fps16 / resident_value, 3), "h2d_bytes_per_frame": int(h2d16),
                              "h2d_gbs": round(h2d16 * fps16 / 1e9, 2),
                              "note": "the same passes with the depth images as raw uint16 (5000 per metre) through msl_sf_fuse_resident_batch_d16: converted on "
                                      "the device as float(raw) * factor (src/Frame.cc:96-97), 2 instead of 4 depth bytes per pixel over PCIe"}
    if do_sf and do_orb and kfe == 1:
        # ONE upload of the gray image for both consumers (msl_sf_staged_gray + msl_orb_wait_event, round 6): the surfel thread enqueues its host-image batch,
        # hands the staged device images and their "uploaded" event to the ORB thread of that batch, and lets the extraction of batch k return before it
        # enqueues batch k + 2 (the staging set of batch k is then overwritten)
        import queue

        def run_shared(npass):
            qs = [queue.Queue() for _ in orbs]
            done = [threading.Event() for _ in range(npass * nsub)]

            def sf_w():
                i = 0
                for _ in range(npass):
                    sf.map_restore()
                    for sb in range(nsub):
                        if i >= 2:
                            done[i - 2].wait()
                        sf.fuse_resident_batch(np.arange(sb * nkf, (sb + 1) * nkf), g_np[sb * B:], d16_np[sb * B:], m_np, kf_poses[sb],
                                               device=False, member_shared=True, frame_step=kfe, depth_factor=d16_factor)
                        qs[sb % len(orbs)].put((i, sb) + sf.staged_gray())
                        i += 1
                sf.sync()
                for q in qs:
                    q.put(None)

            def orb_w(which):
                ex = orbs[which]
                while True:
                    item = qs[which].get()
                    if item is None:
                        return
                    i, sb, p, rs, fs, ev = item
                    ex.wait_event(ev)
                    ex.extract_batch_shared(p, rs, fs, h_kps[sb * B * cap * 28:], h_desc[sb * B * cap * 32:], h_n[sb * B:], B, W, H)
                    done[i].set()

            th = [threading.Thread(target=sf_w)] + [threading.Thread(target=orb_w, args=(i,)) for i in range(len(orbs))]
            for t in th:
                t.start()
            for t in th:
                t.join()

        h_n[:] = 0
        run_shared(2)
        t0 = time.perf_counter()
        run_shared(npass)
        fps_sh = npass * F / (time.perf_counter() - t0)
        h2d_sh = W * H + 2 * W * H
        res["raw_depth16_shared_gray"] = {"value_streaming": round(fps_sh, 1), "fraction_of_resident": round(fps_sh / resident_value, 3), "h2d_bytes_per_frame": int(h2d_sh),
                                          "h2d_gbs": round(h2d_sh * fps_sh / 1e9, 2), "keypoints_per_frame": round(int(h_n.sum()) / F, 1),
                                          "note": "raw 16-bit depth and ONE upload of the gray image for both handles: the ORB extractor reads the images the surfel "
                                                  "batch staged on the device (msl_sf_staged_gray + msl_orb_wait_event + msl_orb_extract_batch with MSL_MEM_DEVICE input)"}
    for ex in orbs:
        ex.close()
    if do_sf:
        sf.close()
    return res


def dropin_shapes(grays, depths, member, poses, smap, W, H, intr, do_orb, do_sf, device):
    """The call shapes the reference's unchanged call sites use (adapter/ORBextractor.cc, adapter/SurfelFusion.cpp): synchronous,
    host buffers in and out, PCIe inclusive.  Never part of `value`."""
    from manhattanslam_amd import ORBextractor, SurfelFusion
    out = {}
    if do_orb:
        ex = ORBextractor(1000, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=1, device=device)
        ex(grays[0])
        t0 = time.perf_counter()
        n = 20
        for i in range(n):
            ex(grays[i % len(grays)])
        out["msl_orb_extract_ms"] = round((time.perf_counter() - t0) * 1e3 / n, 3)
        ex.close()
    if do_sf:
        s2 = SurfelFusion(W, H, intr["fx"], intr["fy"], intr["cx"], intr["cy"], 30.0, 0.5, device=device)
        local = smap.copy()
        s2.fuseInitializeMap(0, grays[0], depths[0], member, poses[0], local)
        t0 = time.perf_counter()
        n = 3
        for i in range(n):
            s2.fuseInitializeMap(1 + i, grays[(1 + i) % len(grays)], depths[(1 + i) % len(grays)], member, poses[(1 + i) % len(grays)], local)
        out["msl_sf_fuse_host_vector_ms"] = round((time.perf_counter() - t0) * 1e3 / n, 3)
        out["msl_sf_fuse_host_vector_surfels"] = len(local)
        t0 = time.perf_counter()
        for i in range(n):   # the same calls with MSL_SF_LOCAL_UNCHANGED: nobody touched `local` since the previous call, so nothing is uploaded
            s2.fuseInitializeMap(4 + i, grays[(4 + i) % len(grays)], depths[(4 + i) % len(grays)], member, poses[(4 + i) % len(grays)], local, local_unchanged=True)
        out["msl_sf_fuse_host_vector_unchanged_hint_ms"] = round((time.perf_counter() - t0) * 1e3 / n, 3)
        s2.close()
    out["note"] = "synchronous drop-in entry points with host buffers (PCIe in and out included); one frame / one keyframe per call"
    return out


if __name__ == "__main__":
    main()
