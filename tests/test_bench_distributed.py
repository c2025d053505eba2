"""The N>1 path of bench.py on CPU (world_size 2 and 4 over gloo), through bench.main's own code path:
`--rehearse` runs the launcher, the rendezvous, the sampler / trainer roles, the DistShuffler step ranges, the REAL
shared hand-off ring of c_lib.so between the rank processes (named shared-memory regions) and the MAX-time / SUM-work
reductions with empty batches -- everything of the multi-process job except the GPU work.  There is no data-path
collective to test beyond this: the path shards by mini-batch."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _check_windows(out, gpus, S, warmup, steps):
    """the steady-state rule of the N >= 2 line: R back-to-back windows of `steps` consecutively consumed batches out of
    ONE span with no barrier inside; value / ms_per_step / edges_per_step belong to the median window"""
    import bench
    T = gpus - S
    wd = out["windows"]
    lead, tail = bench.span_margins(warmup, T)
    assert wd["count"] == 5 and len(wd["ms_per_step"]) == 5 and all(x > 0 for x in wd["ms_per_step"])
    assert wd["lead_batches"] == lead and wd["tail_batches"] == tail and wd["span_batches"] == lead + 5 * steps + tail
    assert out["ms_per_step"] == sorted(wd["ms_per_step"])[2] == wd["ms_per_step"][wd["median_index"]]
    assert wd["min"] <= out["ms_per_step"] <= wd["max"]
    keys = wd["median_window_keys"]
    assert len(keys) == len(set(keys)) == steps
    # every key is a step some sampler owns (its range of the epoch, dist_shuffler.cc:59-79), taken once
    owned = set()
    for i in range(S):
        first, local = bench.local_step_range(151, i, S)
        n = bench.split_count(wd["span_batches"], S, i)
        owned |= {(j // local) * 151 + first + j % local for j in range(n)}
    assert set(keys) <= owned
    assert round(out["edges_per_step"] * steps) == sum(bench.RehearsalBackend.edges_of(k) for k in keys)
    assert abs(out["value"] - out["edges_per_step"] / (out["ms_per_step"] * 1e-3)) <= 1e-6 * out["value"]


def _run(cmd, env=None):
    res = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert res.returncode == 0, res.stderr.decode()[-2000:]
    lines = [ln for ln in res.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout.decode()
    return json.loads(lines[0])


@pytest.mark.parametrize("gpus,samplers", [(2, 0), (4, 2), (3, 0), (8, 0)])
def test_plain_launch_spawns_ranks(gpus, samplers):
    """`python bench.py --gpus N` (how the driver ran it in round 1): bench.py starts the N rank processes itself"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    before = set(os.listdir("/dev/shm"))
    cmd = [sys.executable, "bench.py", "--gpus", str(gpus), "--steps", "20", "--warmup", "5", "--rehearse"]
    if samplers:
        cmd += ["--samplers", str(samplers)]
    out = _run(cmd, env)
    S = samplers or max(1, gpus // 4)  # bench.default_samplers: 2S+6T at 8 GPUs (exp/table4/run.py:329-330)
    assert out["n_gpus"] == gpus and out["steps"] == 20 and out["warmup"] == 5
    assert out["pipeline"]["samplers"] == S and out["pipeline"]["trainers"] == gpus - S
    assert out["config"]["parallelism"].startswith("%dS+%dT" % (S, gpus - S))
    _check_windows(out, gpus, S, 5, 20)
    assert out["pipeline"]["n1_point_of_this_curve"]["value"] is None  # a number on a GPU box, a reason here
    assert out["input_nodes_per_step"] == 1.0  # every one of the 20 batches reached exactly one trainer
    assert out["scaling"] == "strong" and out["ms_per_step"] > 0
    assert out["pipeline"]["handoff"]["transport"] == "none (rehearsal)" and len(out["pipeline"]["handoff"]["rings"]) == S
    # the link self-test's record is part of every N >= 2 line; a rehearsal has no GPU and says so instead of a number
    assert out["pipeline"]["links"]["rccl_world"] is None and "rehearsal" in out["pipeline"]["links"]["why"]
    assert not [f for f in set(os.listdir("/dev/shm")) - before if f.startswith("fgnn_bench_")]  # rank 0 cleaned up


def test_sampler_count_is_chosen_from_measured_rates():
    """--samplers auto: S = argmin max(t_s / S, t_t / (N - S)) over the two per-process rates measured before the roles
    are given out (bench.choose_samplers / calibrate_roles; the reference tunes S per workload by hand,
    exp/table4/README.md:79-90)"""
    import bench
    # this build's rates at presample_epoch 1 (sampler 0.10, trainer 0.29 ms per batch): the reference's 2S+6T balances
    assert [bench.choose_samplers(n, 0.10, 0.29)[0] for n in (3, 4, 8)] == [1, 1, 2]
    # a trainer twice as fast (a three-epoch ranking): a third sampler pays at 8 ranks, a second one at 4
    assert [bench.choose_samplers(n, 0.10, 0.15)[0] for n in (3, 4, 8)] == [1, 2, 3]
    best, pred = bench.choose_samplers(8, 0.10, 0.15)
    assert pred[best] == min(pred.values()) and set(pred) == set(range(1, 8))
    assert bench.choose_samplers(8, 1.0, 0.01)[0] == 7 and bench.choose_samplers(8, 0.01, 1.0)[0] == 1
    assert bench.choose_samplers(4, 0.2, 0.2)[0] == 2  # (a tie between 1 and 3 never beats the balanced split)


@pytest.mark.parametrize("gpus,rates,want", [(8, "0.1,0.29", 2), (8, "0.1,0.15", 3), (4, "0.1,0.15", 2), (2, "0.1,0.15", 1)])
def test_rehearsal_line_shows_the_sampler_choice(gpus, rates, want):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    out = _run([sys.executable, "bench.py", "--gpus", str(gpus), "--steps", "10", "--warmup", "2", "--rehearse",
                "--rehearse-rates", rates], env)
    ch = out["pipeline"]["sampler_choice"]
    assert out["pipeline"]["samplers"] == want and out["pipeline"]["trainers"] == gpus - want
    if gpus >= 3:
        ts, tt = [float(x) for x in rates.split(",")]
        assert ch["mode"] == "auto" and ch["chosen"] == want and ch["sampler_ms_per_batch_alone"] == ts
        assert ch["trainer_ms_per_batch_alone"] == tt and ch["reference_split"] == max(1, gpus // 4)
        assert min(ch["predicted_ms_per_batch_by_samplers"].values()) == ch["predicted_ms_per_batch_by_samplers"][str(want)]
    else:
        assert "no choice" in ch["mode"]
    _check_windows(out, gpus, want, 2, 10)


def test_sampler_keys_of_the_run_config_per_workload():
    """bench.py --gpus N hands the engine the reference's config keys for the workload's sampler: fan-outs for the k-hop
    samplers, the walk parameters for PinSAGE's random walks (operation.cc:58-164)"""
    import bench
    assert bench.sampler_config_keys(bench.WORKLOADS["papers100M"], "khop2") == {"num_fanout": 2, "fanout": [25, 10]}
    assert bench.sampler_config_keys(bench.WORKLOADS["twitter"], "weighted_khop_prefix") == {"num_fanout": 3, "fanout": [5, 10, 15]}
    assert bench.sampler_config_keys(bench.WORKLOADS["uk-2006-05"], "random_walk") == {
        "random_walk_length": 3, "random_walk_restart_prob": 0.5, "num_random_walk": 25, "num_neighbor": 5, "num_layer": 3}
    assert bench.sampler_config_keys(bench.WORKLOADS["small"], "random_walk")["num_random_walk"] == 4  # the reference's default


def test_pipeline_half_is_importable_without_the_entry_point():
    """bench.py is the entry point only; the N >= 2 half (benchlib/pipeline.py) and the N = 1 half import and expose their
    pieces on their own, and bench re-exports the names the tools use"""
    import importlib
    pl = importlib.import_module("benchlib.pipeline")
    assert pl.choose_samplers(8, 0.1, 0.29)[0] == 2 and pl.pipeline_roles(8, "auto") == (2, 6)
    assert pl.span_margins(5, 6) == (340, 12)
    sg = importlib.import_module("benchlib.single")
    assert callable(sg.run_single) and callable(sg.run_extract_leg)
    import bench
    assert bench.choose_samplers is pl.choose_samplers and bench.WORKLOADS is pl.WORKLOADS
    assert len(open(bench.BENCH_PY).read().splitlines()) < 300


def test_torchrun_launch():
    """the launch line of the task description: one rank per GPU started by torch.distributed.run"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    out = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                "127.0.0.1", "--master-port", str(_free_port()), "bench.py", "--gpus", "2", "--steps", "12",
                "--warmup", "3", "--rehearse"], env)
    assert out["n_gpus"] == 2 and out["steps"] == 12
    _check_windows(out, 2, 1, 3, 12)


def test_gpus_flag_must_match_the_launcher():
    env = dict(os.environ, RANK="0", WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(_free_port()))
    res = subprocess.run([sys.executable, "bench.py", "--gpus", "3", "--rehearse"], cwd=ROOT, env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert res.returncode != 0 and b"--gpus 3" in res.stderr


def test_roles_and_shares():
    import bench
    assert [bench.default_samplers(n) for n in (2, 3, 4, 8)] == [1, 1, 1, 2]  # 1S+1T at 2, 2S+6T at 8 (table4/run.py)
    assert bench.pipeline_roles(8) == (2, 6) and bench.pipeline_roles(4, 2) == (2, 2)
    with pytest.raises(ValueError):
        bench.pipeline_roles(2, 2)
    for total in (0, 1, 20, 151):
        for parts in (1, 2, 3, 6):
            assert sum(bench.split_count(total, parts, i) for i in range(parts)) == total


def test_training_region_gives_every_trainer_the_same_number_of_batches():
    """gradient all-reduce per step: an uneven share deadlocks the trainers (seen with 1S+3T: 40 batches = 14+13+13)"""
    import bench
    for steps, train_steps in ((151, 40), (20, 40), (5, 40), (1, 40), (60, 20)):
        for trainers in (1, 2, 3, 5, 6, 7):
            warm, timed = bench.train_region_batches(steps, train_steps, trainers)
            assert warm % trainers == 0 and timed % trainers == 0 and warm >= trainers and timed >= trainers
            assert {bench.split_count(timed, trainers, i) for i in range(trainers)} == {timed // trainers}
            assert timed <= max(min(steps, train_steps), trainers)


def test_span_arithmetic_and_window_reader():
    import bench
    assert bench.span_margins(5, 6) == (340, 12) and bench.span_margins(5, 1) == (340, 5) and bench.span_margins(0, 1) == (340, 2)
    assert bench.span_margins(5, 6, decoupled=True) == (1, 0)
    for T in (1, 2, 3, 6, 7):
        lead, tail = bench.span_margins(5, T)
        assert bench.span_total(lead, 5, 20, tail, T, False) == lead + 100 + tail
        tot = bench.span_total(lead, 5, 18, tail, T, True)
        assert tot % T == 0 and 0 <= tot - (lead + 90 + tail) < T
    # stamps from three trainers, shuffled: batch b is consumed at time 10 + 2 b
    import random
    stamps = [(10.0 + 2.0 * b, 1000 + b) for b in range(3 + 4 * 5 + 2)]
    random.Random(1).shuffle(stamps)
    merged, wins = bench.read_windows(stamps, 3, 4, 5)
    assert [k for _, k in merged] == list(range(1000, 1025))
    for j, (t0, t1, keys) in enumerate(wins):
        assert keys == list(range(1003 + 5 * j, 1008 + 5 * j)) and abs((t1 - t0) - 10.0) < 1e-9
        assert t0 == 10.0 + 2.0 * (3 + 5 * j - 1)  # the clock of a window starts at the batch consumed before it


def _limited_collective_rank(rank, port, q):
    import datetime
    import torch.distributed as dist
    import bench
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE="2")
    dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=60))
    grp = dist.new_group(ranks=[0, 1], backend="gloo", timeout=datetime.timedelta(seconds=20))
    import torch

    def body():
        if rank == 1:
            raise RuntimeError("this rank fails alone")
        t = torch.ones(1)
        dist.all_reduce(t, group=grp)  # rank 1 never joins: stuck until the group's own timeout
        return {"sum": float(t)}
    import time
    t0 = time.time()
    res, bad = bench.limited_collective(dist, 2, body, 2.0)
    q.put((rank, res, bad, time.time() - t0))
    dist.barrier()  # the job's own group still works on both ranks
    os._exit(0)  # rank 0's helper thread is still inside the abandoned collective


def test_a_rank_failing_a_collective_alone_does_not_hang_the_others():
    """bench.link_selftest's rule: the RCCL calls run on a helper thread with a per-rank limit and
    the ranks agree over gloo on who finished -- checked here with a gloo sub-group standing in for RCCL"""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_limited_collective_rank, args=(r, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(2))
    for p in ps:
        p.join(60)
    for rank, res, bad, took in got:
        assert res is None and took < 15.0
        assert set(bad) == {0, 1} and "fails alone" in bad[1] and "no answer" in bad[0]


def test_step_ranges_cover_epoch():
    import bench
    assert bench.local_step_range(151, 0, 2) == (0, 75) and bench.local_step_range(151, 1, 2) == (75, 76)
    for steps in (1, 7, 151, 152):
        for world in (1, 2, 3, 4, 8):
            seen = []
            for r in range(world):
                f, c = bench.local_step_range(steps, r, world)
                seen += list(range(f, f + c))
            assert seen == list(range(steps))


def test_numa_helpers_on_this_host():
    """bench.py's NUMA diagnostics (extract leg, N >= 2 line): nodes with memory from sysfs, page placement through
    move_pages in query mode -- on whatever host this runs (one node is fine)."""
    import ctypes
    import numpy as np
    sys.path.insert(0, ROOT)
    import bench
    nodes = bench.numa_nodes_with_memory()
    assert nodes and nodes[0] == 0 and nodes == sorted(nodes)
    a = np.ones(1 << 20, dtype=np.uint8)  # touched: the pages exist
    hist = bench.pages_by_numa_node(a.ctypes.data, a.nbytes, samples=64)
    assert hist is None or (sum(hist.values()) == 64 and all(int(k) in nodes or int(k) < 0 for k in hist))
    assert bench.gpu_numa_node(0) is None or isinstance(bench.gpu_numa_node(0), int)
