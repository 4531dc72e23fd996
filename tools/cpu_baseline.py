#!/usr/bin/env python3
"""CPU baseline of bench.py: the oracle restatement (oracle/libmsl_oracle.so, g++ -O3, no -march=native --
the reference itself needs OpenCV/Eigen and cannot be built) timed on this host's cores on a bounded sample of the SAME
synthetic workload.  Runs in its own interpreter (no HIP runtime in the process that forks workers) and prints one JSON object:

  single_thread      ORB 1 thread per frame (src/Frame.cc:100) + SurfelFusion on one thread, frame after frame
  surfel_10_threads  SurfelFusion with the reference's THREAD_NUM = 10 fork/join per stage (include/SurfelFusion.h:34)
  throughput         one independent sequence per host core (P worker processes, each ORB + SurfelFusion, single-threaded)

Test infrastructure / reported baseline only; nothing here is on the product path.
"""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def build_inputs(F, n_surfels, W, H, intr_name, rank=0, map_kind="dense", map_order="creation", scene="room"):
    import importlib.util
    spec = importlib.util.spec_from_file_location("msl_synth", os.path.join(ROOT, "manhattanslam_amd", "synth.py"))
    synth = importlib.util.module_from_spec(spec)   # synth.py alone: importing the package would load libmsl.so / HIP
    spec.loader.exec_module(synth)
    intr = synth.scaled_intrinsics(getattr(synth, intr_name), W)
    grays, depths, member, poses, smap = synth.bench_inputs(rank, F, n_surfels, W, H, intr, map_kind=map_kind, map_order=map_order, scene=scene)
    return grays, depths, member, poses, smap, intr


def run_sequence(inp, W, H, n_frames, do_orb=True, do_sf=True, threads10=False, keyframe_every=1):
    from tests import oracle_lib
    grays, depths, member, poses, smap, intr = inp
    o = oracle_lib.load()
    ex = o.orb_create(1000, 1.2, 8, 20, 7) if do_orb else None
    sf = None
    if do_sf:
        sf = oracle_lib.OracleSurfel(W, H, intr["fx"], intr["fy"], intr["cx"], intr["cy"], 30.0, 0.5)
        sf.map_set(smap.astype(oracle_lib.SURFEL_DTYPE))
        if threads10:
            sf.set_threads(True)
    F = len(grays)
    t_orb = t_sf = 0.0
    n_kf = 0
    t_begin = time.perf_counter()
    for i in range(n_frames):
        f = i % F
        t0 = time.perf_counter()
        if ex is not None:
            ex.extract(grays[f])
        t1 = time.perf_counter()
        if sf is not None and i % keyframe_every == 0:
            sf.fuse_map(i, grays[f], depths[f], member, poses[f])
            n_kf += 1
        t2 = time.perf_counter()
        t_orb += t1 - t0
        t_sf += t2 - t1
    return t_begin, time.perf_counter(), t_orb, t_sf, n_kf


def parity_gate(path):
    """SURVEY.md 8(d), last row: the checker half of bench.py's parity gate.  `path` is an .npz bench.py wrote after its timed region: the inputs of
    a few frames / keyframes of the run's OWN workload and what the HIP library made of them (keypoints + descriptors per frame; the resident map
    after the keyframes, starting from the pre-seeded map).  The oracle processes the same inputs; ORB must agree byte for byte (count, every
    cv::KeyPoint field, order, 32 descriptor bytes), the map in length, order and integer fields exactly and in every float field within 1e-4
    (NaNs in the same places).  Prints one JSON object; bench.py exits non-zero when "ok" is false."""
    from tests import oracle_lib
    z = np.load(path, allow_pickle=False)
    W, H = int(z["size"][0]), int(z["size"][1])
    out = {"orb_frames": 0, "keyframes": 0, "ok": True, "max_abs": 0.0, "tolerance": 1e-4, "checker": "oracle/libmsl_oracle.so (parity unpinned: DESIGN.md 3)"}
    fail = []
    o = oracle_lib.load()
    if "orb_gray" in z:
        ex = o.orb_create(1000, 1.2, 8, 20, 7)
        n_kp = 0
        for f, img in enumerate(z["orb_gray"]):
            ko, do = ex.extract(img)
            n = int(z["orb_n"][f])
            kg = z["orb_kps"][f][:n * 28].tobytes()
            dg = z["orb_desc"][f][:n * 32].tobytes()
            if n != len(ko) or kg != ko.tobytes() or dg != do.tobytes():
                fail.append(f"ORB frame {f}: {n} keypoints on the GPU, {len(ko)} in the oracle" + ("" if n != len(ko) else ", bytes differ"))
            n_kp += len(ko)
        out["orb_frames"] = int(len(z["orb_gray"])); out["orb_keypoints"] = n_kp
    if "sf_map_gpu" in z:
        I = z["intr"]
        sf = oracle_lib.OracleSurfel(W, H, float(I[0]), float(I[1]), float(I[2]), float(I[3]), 30.0, 0.5)
        sf.map_set(z["sf_map0"].view(oracle_lib.SURFEL_DTYPE).reshape(-1))
        nk = len(z["sf_refs"])
        for k in range(nk):
            mem = z["sf_member"][k if z["sf_member"].shape[0] > 1 else 0]
            sf.fuse_map(int(z["sf_refs"][k]), np.ascontiguousarray(z["sf_gray"][k]), np.ascontiguousarray(z["sf_depth"][k]), np.ascontiguousarray(mem),
                        np.ascontiguousarray(z["sf_poses"][k], np.float32))
        mo = sf.map_get()
        mg = z["sf_map_gpu"].view(oracle_lib.SURFEL_DTYPE).reshape(-1)
        out["keyframes"] = int(nk); out["surfels"] = int(len(mo)); out["surfels_updated"] = int((mo["lastUpdate"] >= int(z["sf_refs"][0])).sum())
        if len(mg) != len(mo):
            fail.append(f"map: {len(mg)} surfels on the GPU, {len(mo)} in the oracle")
        else:
            for f in ("r", "g", "b", "updateTimes", "lastUpdate"):
                if not np.array_equal(mg[f], mo[f]):
                    fail.append(f"map field {f}: {int((mg[f] != mo[f]).sum())} surfels differ (first at {int(np.flatnonzero(mg[f] != mo[f])[0])})")
            for f in ("px", "py", "pz", "nx", "ny", "nz", "size", "color", "weight"):
                a, b = mg[f].astype(np.float64), mo[f].astype(np.float64)
                if not np.array_equal(np.isnan(a), np.isnan(b)):
                    fail.append(f"map field {f}: NaNs in different places")
                    continue
                d = np.abs(a - b); d = d[~np.isnan(d)]
                m = float(d.max()) if d.size else 0.0
                out["max_abs"] = max(out["max_abs"], m)
                if m > 1e-4:
                    fail.append(f"map field {f}: max |GPU - oracle| = {m:.3g}")
            out["bit_identical"] = bool(mg.tobytes() == mo.tobytes())
    out["ok"] = not fail
    if fail:
        out["failures"] = fail[:8]
    print(json.dumps(out), flush=True)


def _worker(inp, W, H, n_frames, do_orb, do_sf, kfe, barrier, q):
    try:
        barrier.wait()
        q.put(run_sequence(inp, W, H, n_frames, do_orb, do_sf, False, kfe))
    except Exception as e:  # noqa: BLE001
        q.put(("error", repr(e)))


def mem_available_bytes():
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                return int(line.split()[1]) * 1024
    except OSError:
        pass
    return 8 << 30


def usable_cpus():
    """Logical CPUs limited by the affinity mask and the cgroup CPU quota (cpu.max): more runnable processes than that are only throttled."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, -(-int(q) // int(p))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and p > 0:
                n = min(n, max(1, -(-q // p)))
        except (OSError, ValueError):
            pass
    return n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=48, help="frames of the single-thread sample")
    ap.add_argument("--frames-10t", type=int, default=32, help="keyframes of the 10-thread SurfelFusion sample")
    ap.add_argument("--frames-per-proc", type=int, default=0, help="frames each process of the throughput run handles (0 = 16, negative = no throughput run)")
    ap.add_argument("--procs", type=int, default=0, help="processes of the throughput run (0 = one per usable CPU, memory permitting)")
    ap.add_argument("--surfels", type=int, default=1_000_000)
    ap.add_argument("--size", default="640x480")
    ap.add_argument("--intrinsics", default="TUM1")
    ap.add_argument("--distinct-frames", type=int, default=16)
    ap.add_argument("--no-orb", action="store_true")
    ap.add_argument("--no-surfel", action="store_true")
    ap.add_argument("--keyframe-every", type=int, default=1)
    ap.add_argument("--map", default="dense", choices=["dense", "sparse", "moving"])
    ap.add_argument("--map-order", default="creation", choices=["creation", "random"])
    ap.add_argument("--scene", default="room", choices=["room", "clutter"])
    ap.add_argument("--parity-gate", default=None, help="checker half of bench.py's parity gate: the .npz of inputs and HIP results to check against the oracle")
    args = ap.parse_args()
    if args.parity_gate:
        return parity_gate(args.parity_gate)
    W, H = (int(v) for v in args.size.lower().split("x"))
    do_orb, do_sf, kfe = not args.no_orb, not args.no_surfel, args.keyframe_every
    inp = build_inputs(args.distinct_frames, args.surfels if do_sf else 16, W, H, args.intrinsics, map_kind=args.map, map_order=args.map_order, scene=args.scene)
    ncpu = usable_cpus()
    out = {"host_cpus": os.cpu_count() or 1, "usable_cpus": ncpu, "kind": "port",
           "code": "oracle/libmsl_oracle.so (CPU restatement of src/ORBextractor.cc + src/SurfelFusion.cpp + SurfelMapping::fuseMap; g++ -O3, no -march=native)"}
    what = ("ORB + " if do_orb else "") + (f"SurfelFusion every {kfe} frame(s), {args.surfels} seeded surfels ({args.map} map, {args.scene} scene)" if do_sf else "no surfel stage")

    if args.frames > 0:
        tb, te, t_orb, t_sf, nkf = run_sequence(inp, W, H, args.frames, do_orb, do_sf, False, kfe)
        out["single_thread"] = {"value": round(args.frames / (te - tb), 3), "unit": "frames/s", "cores": 1,
                                "sample": f"{args.frames} frames, {what}, {W}x{H}",
                                "orb_ms_per_frame": round(1e3 * t_orb / args.frames, 2),
                                "surfel_ms_per_keyframe": round(1e3 * t_sf / max(nkf, 1), 2)}
    if args.frames_10t > 0 and do_sf:
        tb, te, _, t_sf, nkf = run_sequence(inp, W, H, args.frames_10t, False, True, True, 1)
        out["surfel_10_threads"] = {"value": round(nkf / t_sf, 3), "unit": "keyframes/s", "cores": 10,
                                    "sample": f"{nkf} keyframes, SurfelFusion only with the reference's THREAD_NUM=10 fork/join per stage, "
                                              f"{args.surfels} seeded surfels, {W}x{H}",
                                    "surfel_ms_per_keyframe": round(1e3 * t_sf / max(nkf, 1), 2)}
    if args.frames_per_proc >= 0:
        per_proc = 96 * W * H * 3 + 2 * 56 * (args.surfels if do_sf else 0) + (64 << 20)   # oracle scratch + map + copy + interpreter
        P = args.procs or max(1, min(ncpu, int(0.5 * mem_available_bytes() // per_proc)))
        # bounded sample: about 15 s of CPU work per process at ~0.11 s per frame (every process handles the same number of frames)
        fpp = args.frames_per_proc or 16
        ctx = mp.get_context("fork")
        barrier = ctx.Barrier(P)
        q = ctx.Queue()
        procs = [ctx.Process(target=_worker, args=(inp, W, H, fpp, do_orb, do_sf, kfe, barrier, q)) for _ in range(P)]
        for p in procs:
            p.start()
        res = [q.get() for _ in procs]
        for p in procs:
            p.join()
        err = [r for r in res if r[0] == "error"]
        if err:
            out["throughput"] = {"error": err[0][1]}
        else:
            wall = max(r[1] for r in res) - min(r[0] for r in res)
            out["throughput"] = {"value": round(P * fpp / wall, 2), "unit": "frames/s", "cores": P,
                                 "sample": f"{P} independent sequences (one single-threaded process per CPU this container may use: affinity mask and cgroup cpu.max) x {fpp} frames, "
                                           f"{what}, {W}x{H}",
                                 "wall_s": round(wall, 3)}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
