#!/usr/bin/env python3
"""bench.py -- sampled-edges/sec of the sampling-and-extraction hot path on MI355X.

One "step" = one mini-batch through the whole path, inputs resident in HBM when the timed region starts:
    batch slice of the shuffled train set -> k-hop sampling (khop2) -> dedup/compaction -> remap
    -> cache-index split -> [hand-off to the trainer GPU] -> feature gather + label gather.
Workload (BASELINE.json metric): papers100M-shaped graph (N=111 059 956, E=1 615 685 872, D=128 f32), GraphSAGE fanout
[25,10], batch 8000.  The graph is the R-MAT graph of SURVEY.md 8(d) (fgnn_hip/rmat.py; --graph powerlaw = round 1's).

--gpus 1   one GPU does both halves through the kernel-level C ABI (libfgnn_hip.so), full feature table in HBM; after
           the timed region two more are measured and reported beside it: the sampler-side stage alone, and BASELINE
           config 3's extract leg (features in host memory, HBM cache of the top cache_ratio*N rows ranked by the
           pre-sampler, misses gathered over the host link).
--gpus N   the factored pipeline of the reference (multi_gpu/train_graphsage.py:103-432): S sampler processes and
           N - S trainer processes, one per GPU, driving arch5 through samgraph.torch / c_lib.so -- DistShuffler step
           ranges per sampler (dist_shuffler.cc:59-79), hand-off through the HBM message ring, trainers extracting
           with the pre-sample cache.  No collective on the data path; gloo carries two barriers and the reductions.
           Started by torchrun (RANK / WORLD_SIZE in the environment), or -- when they are absent -- bench.py itself
           starts N fresh rank processes before it touches the GPU and relays rank 0's line.
The timed region covers K steps IN TOTAL for every N ("scaling": "strong").

Prints ONE JSON line on rank 0 (contract in the task description).

The code lives in benchlib/: common (inputs, constants), single (N = 1), pipeline (N >= 2), cpu (the CPU baselines); this
file is the entry point -- arguments, the rank launcher, the dispatch -- and re-exports their names (`bench.X`).
"""
import argparse
import json
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from benchlib.common import *  # noqa: E402,F401,F403
from benchlib.common import BENCH_PY, ROOT, WORKLOADS, SAMPLE_TYPES, lib, torch  # noqa: E402,F401
from benchlib.cpu import *  # noqa: E402,F401,F403
from benchlib.single import *  # noqa: E402,F401,F403
from benchlib.pipeline import *  # noqa: E402,F401,F403
from benchlib.cpu import cpu_baseline_products  # noqa: E402
from benchlib.pipeline import _FileBarrier, run_calibrate_child, run_n1_point_child, run_pipeline_rank  # noqa: E402,F401
from benchlib.single import run_single  # noqa: E402


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes (nothing in this process has touched
    the GPU), let them rendezvous on 127.0.0.1, relay rank 0's JSON line."""
    import socket
    n = args.gpus
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), FGNN_BENCH_CHILD="1")
        procs.append(subprocess.Popen([sys.executable, BENCH_PY] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    line = None
    rc = 0
    try:
        out0 = b""
        while True:
            try:
                out0, _ = procs[0].communicate(timeout=1.0)
                break
            except subprocess.TimeoutExpired:
                bad = [p.returncode for p in procs[1:] if p.poll() not in (None, 0)]
                if bad:  # a rank died: the others would wait for it until the rendezvous times out
                    rc = bad[0]
                    break
        for ln in out0.decode(errors="replace").splitlines():
            if ln.startswith("{") and '"metric"' in ln:
                line = ln
        if not rc:
            for p in procs:
                rc = rc or p.wait()
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    if line is None or rc:
        sys.exit("bench.py: rank processes failed (rc %s)" % rc)
    print(line, flush=True)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=151)   # one papers100M epoch at batch 8000
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--windows", type=int, default=5,
                    help="timed windows of --steps steps each, back to back; `value` is the median window's")
    ap.add_argument("--workload", default=os.environ.get("FGNN_BENCH_WORKLOAD", "papers100M"), choices=list(WORKLOADS))
    ap.add_argument("--graph", default=os.environ.get("FGNN_BENCH_GRAPH", "rmat"), choices=["rmat", "powerlaw"],
                    help="rmat: SURVEY.md 8(d)'s generator (default); powerlaw: round 1's locality-free generator")
    ap.add_argument("--cache-ratio", type=float, default=0.2)
    ap.add_argument("--presample-variants", default="3",
                    help="N=1 extract leg: further presample_epoch values measured after the primary one and reported "
                         "under roofline_extract.variants (comma-separated; empty: none)")
    ap.add_argument("--presample-epochs", type=int, default=1,
                    help="RunConfig::presample_epoch of the pre-sample cache policy (dist/pre_sampler.cc:75-162): epochs "
                         "the access frequencies are counted over.  The reference's scripts default to 1 and its experiment "
                         "runner sweeps 1-3 (exp/common/runner_helper.py:47-49).  One epoch touches 0.105 N distinct nodes "
                         "-- half of a 0.2 cache stays unranked: hit rate 0.905 / 0.938 / 0.953 / 0.962 / 0.968 for 1-5 "
                         "epochs on the papers100M shape (profiles/r05_c_presample_epochs.txt), 15 ms of sampling each")
    ap.add_argument("--empty-feat-bits", type=int, default=24,
                    help="host feature table of 2^k rows, node ids masked (SAMGRAPH_EMPTY_FEAT)")
    ap.add_argument("--host-feat-numa", default="auto",
                    help="N=1 extract leg: where the host feature table lives: auto / gpu = the GPU's NUMA node, node:<n>, "
                         "runtime = torch pin_memory")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--eager-train", action="store_true",
                    help="train leg: op-by-op training step instead of the captured HIP graph (examples/graphed_step.py)")
    ap.add_argument("--no-gemm-tuning", action="store_true",
                    help="train leg: capture the step with the GEMM library's default kernel picks (default: "
                         "PyTorch's TunableOp chooses per GEMM shape before a shape is captured, "
                         "examples/graphed_step.py)")
    ap.add_argument("--no-extract-leg", action="store_true", help="N=1: skip the cache-0.2 / host-miss extract leg")
    ap.add_argument("--no-overlap", action="store_true", help="one host thread, one stream, batches back to back")
    ap.add_argument("--timed-only", action="store_true",
                    help="N=1: skip the extra measurements after the timed region: under rocprofv3 --stats every "
                         "gather launch is then one of the overlapped kind the roofline line is computed from")
    ap.add_argument("--host-threads", type=int, default=1, help="N=1: host threads enqueueing batches")
    ap.add_argument("--streams-per-thread", type=int, default=3,
                    help="N=1: HIP streams each host thread rotates over (batches in flight = threads x this); "
                         "measured on MI355X: 1x3 0.160 ms/step, 1x2 = 2x1 0.173, 3x1 0.167-0.177, 1x4 0.183")
    ap.add_argument("--buffers-per-stream", type=int, default=2,
                    help="N=1: batch buffers per stream (the host enqueues that many batches ahead on each stream)")
    ap.add_argument("--stage-streams", type=int, default=2,
                    help="N=1: batch streams of the sample_stage measurement (0: all of --streams-per-thread)")
    ap.add_argument("--samplers", default="auto",
                    help="N>=2: sampler processes; auto (default): chosen from the measured rates of one sampler process "
                         "and one trainer process alone (calibrate_roles) -- the reference tunes it per workload by hand "
                         "(exp/table4: 2 at 8 GPUs for this workload)")
    ap.add_argument("--rehearse-rates", default="0.1,0.29",
                    help="--rehearse with --samplers auto: the (sampler, trainer) ms per batch the choice is made from")
    ap.add_argument("--calibrate-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--no-train-leg", action="store_true", help="skip the region with a training step per batch")
    ap.add_argument("--train-steps", type=int, default=40, help="N>=2: batches of the training region (<= --steps)")
    ap.add_argument("--decoupled", action="store_true",
                    help="N>=2 diagnostic (one-GPU development box, where sampler and trainer ranks share the GPU): the "
                         "samplers fill the queue first, then the trainers drain it -- sampler_busy_s and trainer_busy_s "
                         "are then each stage's time ALONE; needs steps <= queue slots (set "
                         "SAMGRAPH_DEVICE_RING_SLOTS >= steps to keep the payloads in HBM)")
    ap.add_argument("--no-n1-point", action="store_true",
                    help="N>=2: skip the like-for-like N = 1 point (the same pipeline on one GPU, run by a child of rank "
                         "0 after the spans)")
    ap.add_argument("--n1-point-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--rehearse", action="store_true",
                    help="N>=2 without a GPU: launcher, rendezvous, roles, step ranges, the real shared ring and the "
                         "reductions with empty batches (tests); measures nothing")
    ap.add_argument("--sample-type", default=None, choices=list(SAMPLE_TYPES),
                    help="default: the workload's (khop2 = the reference's default for GraphSAGE, "
                         "multi_gpu/train_graphsage.py:75)")
    ap.add_argument("--num-walks", type=int, default=0,
                    help="random_walk workloads: walks per seed (default: the workload's 25; the reference's PinSAGE "
                         "scripts default to 4, multi_gpu/train_pinsage.py:130-134)")
    ap.add_argument("--seed", type=lambda s: int(s, 0), default=0x5A4D47)
    ap.add_argument("--kernel-lib", default=None,
                    help="tools/ only: another build of the kernel library for the N = 1 path ('prof' = "
                         "lib/libfgnn_hip_prof.so, the build that reads FGNN_* A/B switches); the product loads "
                         "lib/libfgnn_hip.so and reads nothing from the environment")
    ap.add_argument("--cpu-only", action="store_true",
                    help="BASELINE.json configs[0] only: the reference's CPU path (oracle/_ref) on the products shape, "
                         "fanout 10/5; needs no GPU and measures nothing of the product")
    return ap.parse_args(argv)


def main():
    # the HIP runtime's hardware queues per process (read at its first call): an arch5 rank runs five streams
    # (samgraph_config sets the same default; here it also covers the torch side of a trainer rank)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # hipIpc handles of the samplers' HBM rings (and RCCL) need it
    args = parse_args()
    if args.kernel_lib:  # measurement tools only: the profiling build of the kernel library
        lib.use_library(lib.PROF_LIB_PATH if args.kernel_lib == "prof" else args.kernel_lib)
    if args.n1_point_child:
        return run_n1_point_child()
    if args.calibrate_child:
        return run_calibrate_child()
    if args.cpu_only:
        dev = torch.device("cuda", 0) if torch.cuda.is_available() else torch.device("cpu")
        r = cpu_baseline_products(dev, budget_s=10.0)
        print(json.dumps({"metric": "sampled-edges/sec (reference CPU path, products fanout 10/5)", "value": r["value"],
                          "unit": "edges/s", "n_gpus": 0, "higher_is_better": True, "data": "synthetic", "dtype": "u32",
                          "config": {"workload": r["config"]}, "cpu_baseline": r}), flush=True)
        return
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None:
        if args.gpus > 1:
            return launch_ranks(args)  # before anything touches the GPU
        return run_single(args)
    world, rank = int(env_world), int(os.environ.get("RANK", "0"))
    if args.gpus not in (1, world):
        sys.exit("bench.py: --gpus %d but the launcher started %d ranks" % (args.gpus, world))
    if world == 1:
        return run_single(args)
    return run_pipeline_rank(args, rank, world)


if __name__ == "__main__":
    main()
