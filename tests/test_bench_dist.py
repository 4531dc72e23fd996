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
