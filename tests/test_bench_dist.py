"""The multi-GPU leg of bench.py (max-over-ranks timing + counter all_gather) with world_size 2 on gloo/CPU."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    dist.barrier()
    local_ms = 10.0 + 5.0 * rank                    # rank 1 is slower: the job time is the max
    counters = [640, 1000 * (rank + 1), 990000 + rank, 3, 50000, 7, int(local_ms * 1e6), 1000000]
    total_ms, gathered = bench.aggregate(local_ms, counters, world, device=None)
    q.put((rank, total_ms, gathered))
    dist.barrier()
    dist.destroy_process_group()


def test_aggregate_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, total_ms, gathered in res:
        assert total_ms == 15.0                      # max over ranks
        assert [g[1] for g in gathered] == [1000, 2000] and [g[2] for g in gathered] == [990000, 990001]
    # whole-job value = frames of all ranks / max time (weak scaling)
    frames = sum(g[0] for g in res[0][2])
    assert frames / (res[0][1] * 1e-3) == 1280 / 0.015


def test_aggregate_single_rank_passthrough():
    sys.path.insert(0, ROOT)
    import bench
    assert bench.aggregate(3.5, [1, 2, 3], 1) == (3.5, [[1, 2, 3]])


def _run_bench(*argv, env=None):
    import json
    import subprocess
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], capture_output=True, text=True, timeout=300, env=e)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r.returncode, (json.loads(lines[-1]) if lines else None), r.stderr


def test_bench_self_launches_its_ranks():
    """`python bench.py --gpus 2` with no rendezvous in the environment must start 2 ranks itself (torch.distributed.run) and
    report n_gpus = 2 with one counter record per rank; --dry-run swaps RCCL for gloo and skips the GPU work."""
    rc, line, err = _run_bench("--gpus", "2", "--dry-run", "--steps", "2", "--frames-per-pass", "8", "--passes-per-step", "1")
    assert rc == 0, err[-2000:]
    assert line["n_gpus"] == 2 and line["dry_run"] is True
    assert len(line["counters_per_rank"]) == 2 and [c[1] for c in line["counters_per_rank"]] == [1000, 2000]
    assert line["value"] == round(2 * 2 * 8 / 0.015, 1)          # frames of all ranks / max-over-ranks time


def test_bench_refuses_gpus_world_mismatch():
    rc, line, err = _run_bench("--gpus", "2", "--dry-run", env={"WORLD_SIZE": "1", "RANK": "0"})
    assert rc != 0 and line is None and "refusing" in err


def test_bench_eight_ranks_bind_distinct_devices_and_sequences():
    """The 8-GPU shape of BASELINE config 5 without hardware: `bench.py --gpus 8 --config 5 --dry-run` starts 8 gloo ranks; every
    rank reports the device it would bind (= its LOCAL_RANK, passed as `device` to both handles, which call hipSetDevice(device) at
    every entry) and the seeds of its own sequence; one 8-record gather reaches rank 0."""
    rc, line, err = _run_bench("--gpus", "8", "--config", "5", "--dry-run", "--steps", "2")
    assert rc == 0, err[-2000:]
    assert line["n_gpus"] == 8 and len(line["counters_per_rank"]) == 8
    b = line["binding_per_rank"]
    assert [x["rank"] for x in b] == list(range(8))
    assert [x["device"] for x in b] == list(range(8)) and all(x["device"] == x["local_rank"] for x in b)
    assert all(x["local_world_size"] == 8 for x in b)
    for k in ("frame_seed", "orb_seed", "map_seed"):
        assert len({x[k] for x in b}) == 8, k                      # independent sequences: seeds offset by rank
    frames = sum(c[0] for c in line["counters_per_rank"])
    assert frames == 8 * 2 * line["config"]["frames_per_pass"] * line["config"]["passes_per_step"]
    assert line["value"] == round(frames / 0.045, 1)               # rank 7 is the slowest fabricated rank: 10 + 5 * 7 ms


def test_eight_ranks_with_two_sequences_per_gpu():
    """--sequences-per-gpu 2 on the 8-GPU shape: rank r runs the global sequences 2 r and 2 r + 1 (16 distinct seed sets over the node), binds
    device r, and the frame count of the gathered records doubles."""
    rc, line, err = _run_bench("--gpus", "8", "--config", "5", "--dry-run", "--steps", "2", "--sequences-per-gpu", "2")
    assert rc == 0, err[-2000:]
    assert line["n_gpus"] == 8 and line["sequences_per_gpu"] == 2
    b = line["binding_per_rank"]
    assert [x["device"] for x in b] == list(range(8)) and all(x["sequences"] == 2 for x in b)
    assert line["global_sequences_per_rank"] == [[2 * r, 2 * r + 1] for r in range(8)]
    assert [x["frame_seed"] for x in b] == [7 + 1000 * 2 * r for r in range(8)]         # first sequence of every rank: no seed shared between ranks
    frames = sum(c[0] for c in line["counters_per_rank"])
    assert frames == 8 * 2 * 2 * line["config"]["frames_per_pass"] * line["config"]["passes_per_step"]


def test_pmc_traffic_is_quoted_only_for_the_profiled_kernel_sources(tmp_path, monkeypatch):
    """roofline.traffic comes from the committed PMC summary only while the kernel sources still hash to the value stored in it."""
    import json
    sys.path.insert(0, ROOT)
    import bench
    got, src = bench.pmc_traffic("k_fuse", "frontend")
    doc = json.load(open(os.path.join(ROOT, "profiles", open(os.path.join(ROOT, "profiles", "current.txt")).read().strip() + "_summary.json")))
    if doc["_meta"]["kernel_source_hash"] == bench.kernel_source_hash():
        assert got == doc["k_fuse"]["fetch_bytes_corrected"] + doc["k_fuse"]["write_bytes"] and src.endswith("_summary.json")
    else:
        assert got is None and "hash mismatch" in src
    monkeypatch.setattr(bench, "kernel_source_hash", lambda: "0" * 16)
    got, src = bench.pmc_traffic("k_fuse", "frontend")
    assert got is None and "hash mismatch" in src
    assert bench.pmc_traffic("k_fuse", "9")[0] is None          # no summary for that configuration
