"""The N>1 path of bench.py on CPU: one process per rank over gloo (world_size 2), disjoint step ranges, MAX-time /
SUM-work reduction.  There is no data-path collective to test beyond this: the path shards by mini-batch."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    first, count = bench.local_step_range(151, rank, world)
    elapsed, edges, rows = bench.reduce_over_ranks(1.0 + rank, 1000.0 * (rank + 1), 10.0 * (rank + 1))
    out.put((rank, first, count, elapsed, edges, rows))
    dist.destroy_process_group()


def test_two_ranks_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 1000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, f0, c0, t0, e0, w0), (r1, f1, c1, t1, e1, w1) = res
    assert (f0, c0, f1, c1) == (0, 75, 75, 76)          # papers100M: 151 steps -> 75 + 76 (dist_shuffler.cc:59-79)
    assert t0 == t1 == 2.0 and e0 == e1 == 3000.0 and w0 == w1 == 30.0


def test_step_ranges_cover_epoch():
    sys.path.insert(0, ROOT)
    import bench
    for steps in (1, 7, 151, 152):
        for world in (1, 2, 3, 4, 8):
            seen = []
            for r in range(world):
                f, c = bench.local_step_range(steps, r, world)
                seen += list(range(f, f + c))
            assert seen == list(range(steps))


def test_single_process_passthrough():
    sys.path.insert(0, ROOT)
    import bench
    assert bench.reduce_over_ranks(1.5, 10, 2) == (1.5, 10, 2)
