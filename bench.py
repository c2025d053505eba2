#!/usr/bin/env python3
"""bench.py -- sampled-edges/sec of the sampling-and-extraction hot path on MI355X.

One "step" = one mini-batch through the whole path on one GPU, inputs resident in HBM:
    batch slice of the shuffled train set -> k-hop sampling (khop2) -> dedup/compaction -> remap
    -> cache-index split -> feature gather + label gather -> batch summary to pinned host memory.
Workload at N=1 (default): papers100M-shaped synthetic graph (BASELINE.json metric: GraphSAGE fanout
[25,10], batch 8000, N=111 059 956, E=1 615 685 872, D=128 f32), full feature table resident in HBM.
For N>1 every rank holds a full replica and samples its own DistShuffler step range
(dist/dist_shuffler.cc:59-79); there is no collective on the data path (weak scaling).

Prints ONE JSON line on rank 0 (contract in the task description).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "fgnn-artifacts_amd"))

from fgnn_hip import lib, synth  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md

WORKLOADS = {
    # name: shape + run config (reference defaults: batch 8000, common_config.py:63; fanout train_graphsage.py:77)
    "papers100M": dict(**synth.DATASET_SHAPES["papers100M"], fanout=[25, 10], batch_size=8000, sample_type="khop2"),
    "products": dict(**synth.DATASET_SHAPES["products"], fanout=[25, 10], batch_size=8000, sample_type="khop2"),
    # BASELINE.json config 4's sampler side: GCN with weighted sampling (multi_gpu/train_gcn.py:72 fanout [5,10,15])
    "twitter": dict(**synth.DATASET_SHAPES["twitter"], fanout=[5, 10, 15], batch_size=8000,
                    sample_type="weighted_khop_prefix"),
    # config 5's sampler side: PinSAGE random walks (multi_gpu/train_pinsage.py:130-134 with num_walks = 25)
    "uk-2006-05": dict(**synth.DATASET_SHAPES["uk-2006-05"], fanout=[5, 5, 5], batch_size=8000,
                       sample_type="random_walk", walk_len=3, num_walks=25, restart_prob=0.5),
    "small": dict(num_node=1_000_000, num_edge=20_000_000, feat_dim=128, num_class=47, num_train=100_000,
                  fanout=[25, 10], batch_size=8000, sample_type="khop2"),
}
SAMPLE_TYPES = {"khop0": lib.KHOP0, "khop1": lib.KHOP1, "khop2": lib.KHOP2, "weighted_khop_prefix": lib.WEIGHTED_KHOP_PREFIX,
                "random_walk": lib.RANDOM_WALK, "weighted_khop": lib.WEIGHTED_KHOP,
                "weighted_khop_hash_dedup": lib.WEIGHTED_KHOP_HASH_DEDUP}


def gen_alias_on_gpu(indices, total, seed, device):
    """prob_table f32[E] / alias_table u32[E] (node ids) for the alias-method samplers: random acceptance
    probabilities, alias = the row neighbour one position further (any node id is a valid table entry) -- same memory
    behaviour as a real table; bit-exact parity with the oracle is covered by the tests, not by the bench."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    prob = torch.empty(total, dtype=torch.float32, device=device)
    chunk = 1 << 27
    for a in range(0, total, chunk):
        prob[a:a + chunk] = torch.rand(min(chunk, total - a), generator=g, device=device)
    alias = torch.roll(indices, 1)
    return prob, alias


def gen_prefix_on_gpu(indptr, total, seed, device):
    """prob_prefix_table (f32[E], per-row inclusive prefix sums of random edge weights), built in row chunks."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    out = torch.empty(total, dtype=torch.float32, device=device)
    ip = indptr.to(torch.int64) & 0xFFFFFFFF
    n = ip.numel() - 1
    rows_per = 1 << 22
    for r0 in range(0, n, rows_per):
        r1 = min(n, r0 + rows_per)
        a, b = int(ip[r0]), int(ip[r1])
        if b == a:
            continue
        wts = torch.rand(b - a, generator=g, device=device, dtype=torch.float32).to(torch.float64) + 1e-3
        cs = torch.cumsum(wts, 0)
        lens = ip[r0 + 1:r1 + 1] - ip[r0:r1]
        starts = ip[r0:r1] - a
        base = torch.where(starts > 0, cs[(starts - 1).clamp_(min=0)], torch.zeros((), dtype=torch.float64, device=device))
        out[a:b] = (cs - torch.repeat_interleave(base, lens)).to(torch.float32)
        del wts, cs, lens, starts, base
    return out


def gen_graph_on_gpu(num_node, num_edge, seed, device):
    """Same construction as synth.powerlaw_csr (power-law row lengths, hub-skewed neighbour ids), done with
    torch on the GPU in chunks so that a 1.6 G-edge CSR is built in seconds without host memory."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    u = torch.rand(num_node, generator=g, device=device, dtype=torch.float64)
    raw = (1.0 - u).pow(-1.0 / 1.8) - 1.0 + 0.05
    raw[torch.rand(num_node, generator=g, device=device) < 0.02] = 0.0
    deg = torch.floor(raw * (num_edge / raw.sum())).to(torch.int64)
    # heavy tail: cap a single row at 2^24 entries and spread the remainder uniformly
    deg.clamp_(max=1 << 24)
    short = int(num_edge - int(deg.sum()))
    if short > 0:
        bump = torch.randint(0, num_node, (short,), generator=g, device=device)
        deg.index_add_(0, bump, torch.ones_like(bump))
    elif short < 0:
        big = torch.nonzero(deg > 0).flatten()
        take = big[torch.randperm(big.numel(), generator=g, device=device)[:(-short)]]
        deg[take] -= 1
    indptr64 = torch.zeros(num_node + 1, dtype=torch.int64, device=device)
    torch.cumsum(deg, 0, out=indptr64[1:])
    total = int(indptr64[-1])
    assert total < 2**32
    indptr = (indptr64 & 0xFFFFFFFF).to(torch.int32) if total >= 2**31 else indptr64.to(torch.int32)
    mul = 2654435761 % num_node
    while np.gcd(mul, num_node) != 1:
        mul += 1
    indices = torch.empty(total, dtype=torch.int32, device=device)
    chunk = 1 << 26
    for a in range(0, total, chunk):
        b = min(total, a + chunk)
        x = torch.rand(b - a, generator=g, device=device, dtype=torch.float64)
        ids = torch.clamp((num_node * x * x).to(torch.int64), max=num_node - 1)
        ids = (ids * mul) % num_node
        indices[a:b] = ids.to(torch.int32)
        del x, ids
    del deg, indptr64, raw, u
    return indptr, indices, total


def gen_features_on_gpu(num_node, dim, device):
    feat = torch.empty((num_node, dim), dtype=torch.float32, device=device)
    rows = max(1, (1 << 28) // dim)
    col = torch.arange(dim, device=device, dtype=torch.int32)[None, :] * 7
    for a in range(0, num_node, rows):
        b = min(num_node, a + rows)
        r = torch.arange(a, b, device=device, dtype=torch.int32)[:, None] * 131
        feat[a:b] = ((r + col) & 0xFFFF).to(torch.float32) * (1.0 / 65536.0)
    return feat


def reduce_over_ranks(elapsed, edges, rows, device=None):
    """Contract: time = MAX over ranks, work = SUM over ranks (no other collective touches the data path)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return elapsed, edges, rows
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    e = torch.tensor([edges, rows], dtype=torch.float64, device=device)
    dist.all_reduce(e, op=dist.ReduceOp.SUM)
    return float(t[0]), float(e[0]), float(e[1])


def local_step_range(steps_per_epoch, rank, world):
    """First step and count of this rank's contiguous step range (DistShuffler, dist/dist_shuffler.cc:59-79)."""
    first = (steps_per_epoch // world) * rank
    count = steps_per_epoch - first if rank == world - 1 else steps_per_epoch // world
    return first, count


def pmc_traffic_ratio():
    """HBM bytes / algorithmic bytes of the feature gather, from the committed rocprofv3 PMC passes
    (profiles/r01_pmc_traffic.json: 2*FETCH_SIZE + WRITE_SIZE, corrected as MI355X_MICROARCH.md prescribes)."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as f:
            return json.load(f)["traffic_over_algorithmic"]
    except Exception:
        return None


def algorithmic_bytes(metas, feat_dim, batch_size):
    """per batch, 4-byte ids.  Returns dict of per-stage algorithmic bytes (sums)."""
    sample = dedup = split = gather = 0
    for m in metas:
        L = m.num_layers
        for l in range(L):
            S, E = m.num_dst[l], m.num_edge[l]
            n_new = m.num_src[l] - m.num_dst[l]
            sample += S * 12 + E * 12
            dedup += E * 16 + n_new * 4
        U = m.num_input
        split += U * 16
        gather += U * (4 + 8 * feat_dim) + m.num_output * 20
    return dict(sample=sample, dedup_remap=dedup, cache_split=split, gather=gather)


def cpu_baseline(w, indptr, indices, feat, train, budget_s=12.0, sample_type="khop2"):
    """The reference's CPU sampling path (CPUSampleKHop0/2 + CPUHashTable2 + CPUExtract driven as DoCPUSample /
    DoFeatureExtract, cpu/cpu_loops.cc:55-227) timed on this host on a bounded number of batches of the same
    workload, multi-threaded (OpenMP, a few thread counts) and single-threaded.  kind "reference": the reference's own
    sources as compiled into oracle/_ref by `make -C oracle _ref` (built files travel with the repo snapshot);
    kind "port": the oracle's restatement of the same functions when oracle/_ref is not there."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py as oracle
    oracle.build()
    t0 = time.time()
    h_indptr = indptr.cpu().numpy().view(np.uint32)
    h_indices = indices.cpu().numpy().view(np.uint32).copy()
    mock_bits = min(24, int(np.floor(np.log2(feat.shape[0]))))
    h_feat = feat[:1 << mock_bits].cpu().numpy()
    copy_s = time.time() - t0
    num_node = h_indptr.shape[0] - 1
    fan, bs = w["fanout"], w["batch_size"]
    cap = oracle.predict_num_nodes(bs, fan)
    h_train = train.cpu().numpy().view(np.uint32)
    mask = (1 << mock_bits) - 1
    out = np.empty((cap, h_feat.shape[1]), dtype=np.float32)
    # thread count: more is not faster for this path (parallel-region and NUMA costs; 16 was best on a 2 x 64-core
    # EPYC 9575F), so a few counts are tried and the best is reported
    if "FGNN_CPU_BASELINE_THREADS" in os.environ:
        cands = [int(os.environ["FGNN_CPU_BASELINE_THREADS"])]
    else:
        cands = sorted({t for t in (8, 16, 32, 64) if t <= (os.cpu_count() or 1)} or {1})
    use_ref = oracle.RefBaseline.available()
    res = {}
    runs = [("omp%d" % t, t) for t in cands] + [("single", 1)]
    budget_s = budget_s / len(runs)
    max_edges = max(bs * int(np.prod([f + 1 for f in fan[i + 1:]])) * fan[i] for i in range(len(fan)))
    ref = None
    if use_ref:
        try:
            ref = oracle.RefBaseline(num_node, max_edges, cap, 1)
        except (OSError, MemoryError, RuntimeError) as e:  # built for another libc / not loadable here: time the port
            print("cpu_baseline: oracle/_ref not usable (%s), timing the oracle's restatement instead" % e, file=sys.stderr)
            use_ref = False
    for label, T in runs:
        if use_ref:
            ref.set_threads(T)
            ctx = None
        else:
            ctx = oracle.OmpBaseline(num_node, cap, T)
        edges = rows = nb = 0
        t_total = 0.0
        warm = 2  # untimed: OpenMP thread-pool start-up and first touch of the tables
        k = 0
        while t_total < budget_s and (k + 1) * bs <= len(h_train):
            seeds = np.ascontiguousarray(h_train[k * bs:(k + 1) * bs])
            t1 = time.time()
            if use_ref:
                e, n_in = ref.sample_batch(h_indptr, h_indices, seeds, fan,
                                           oracle.KHOP2 if sample_type == "khop2" else oracle.KHOP0, h_feat, mock_bits, out)
            else:
                e, n_in = ctx.sample_batch(h_indptr, h_indices, seeds, fan, h_feat, mask, out)
            dt = time.time() - t1
            k += 1
            if k <= warm:
                continue
            t_total += dt
            edges += e
            rows += n_in
            nb += 1
        res[label] = dict(threads=T, batches=nb, seconds=t_total, edges_per_s=edges / t_total, rows_per_s=rows / t_total)
    if ref is not None:
        ref.close()
    best = max(res.values(), key=lambda r: r["edges_per_s"])
    what = ("the reference's own CPU sources (cpu_sampling_khop2.cc, cpu_hashtable2.cc, cpu_extraction.cc, cpu_random.cc "
            "compiled unmodified into oracle/_ref, driven as DoCPUSample / DoFeatureExtract, cpu_loops.cc:55-227)"
            if use_ref else "oracle restatement of CPUSampleKHop2 + CPUHashTable2 + CPUExtract (oracle/_ref not present)")
    return {
        "value": best["edges_per_s"], "unit": "sampled-edges/s", "cores": best["threads"],
        "kind": "reference" if use_ref else "port",
        "sample": f"{best['batches']} batches of {bs} seeds, fanout {fan}, same graph, whole path (sample + dedup + remap "
                  f"+ feature gather) in {best['seconds']:.1f}s with {best['threads']} OpenMP threads; single thread: "
                  f"{res['single']['edges_per_s']:.3e} edges/s; feature table masked to 2^{mock_bits} rows "
                  f"(SAMGRAPH_EMPTY_FEAT / CPUMockExtract); host copy of CSR/features {copy_s:.1f}s not counted; {what}",
        "rows_per_s": best["rows_per_s"], "single_thread_edges_per_s": res["single"]["edges_per_s"],
        "all_runs": res,
        "host_cpus": os.cpu_count(),
    }


def cpu_baseline_generic(w, args, indptr, indices, prefix, feat, train, budget_s=12.0):
    """Weighted / random-walk workloads: the oracle's single-thread restatement of the same pipeline (the reference
    has no CPU twin of these samplers: its arch0 supports khop0/khop2 only, cpu_loops.cc:84-97), a few batches."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py as oracle
    oracle.build()
    h_indptr = indptr.cpu().numpy().view(np.uint32)
    h_indices = indices.cpu().numpy().view(np.uint32).copy()
    h_prefix = prefix.cpu().numpy() if prefix is not None else None
    mock_bits = min(24, int(np.floor(np.log2(feat.shape[0]))))
    h_feat = feat[:1 << mock_bits].cpu().numpy()
    mask = (1 << mock_bits) - 1
    fan, bs = w["fanout"], w["batch_size"]
    num_node = h_indptr.shape[0] - 1
    st = {"weighted_khop_prefix": oracle.WEIGHTED_KHOP_PREFIX, "random_walk": oracle.RANDOM_WALK,
          "khop1": oracle.KHOP1}[args.sample_type]
    kw = {}
    if st == oracle.WEIGHTED_KHOP_PREFIX:
        kw = dict(prob_prefix=h_prefix)
    if st == oracle.RANDOM_WALK:
        kw = dict(walk_len=w["walk_len"], num_walks=w["num_walks"], num_neighbor=fan[0], restart_prob=w["restart_prob"])
    rng = oracle.make_rng(oracle.RNG_PHILOX, args.seed)
    ht = oracle.HashTable(num_node, oracle.predict_num_nodes(bs, fan))
    h_train = train.cpu().numpy().view(np.uint32)
    edges = rows = nb = 0
    t_total = 0.0
    k = 0
    while t_total < budget_s and (k + 1) * bs <= len(h_train):
        seeds = np.ascontiguousarray(h_train[k * bs:(k + 1) * bs])
        t1 = time.time()
        task = oracle.do_sample(h_indptr, h_indices, seeds, fan, st, rng, k, ht, **kw)
        _ = h_feat[task["input_nodes"] & mask]
        t_total += time.time() - t1
        edges += task["total_edges"]
        rows += len(task["input_nodes"])
        nb += 1
        k += 1
    return {"value": edges / t_total, "unit": "sampled-edges/s", "cores": 1, "kind": "port",
            "sample": f"{nb} batches of {bs} seeds, {args.sample_type} fanout {fan}, same graph, whole path (sample + dedup "
                      f"+ remap + feature gather) in {t_total:.1f}s, single thread (oracle restatement; the reference has "
                      f"no CPU twin of this sampler); feature table masked to 2^{mock_bits} rows",
            "rows_per_s": rows / t_total, "host_cpus": os.cpu_count()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=151)   # one papers100M epoch at batch 8000
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default=os.environ.get("FGNN_BENCH_WORKLOAD", "papers100M"), choices=list(WORKLOADS))
    ap.add_argument("--cache-ratio", type=float, default=0.2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-overlap", action="store_true", help="one host thread, one stream, batches back to back")
    ap.add_argument("--timed-only", action="store_true",
                    help="skip the two extra measurements after the timed region (sampler-side stage alone, gather "
                         "alone): under rocprofv3 --stats every gather launch is then one of the overlapped kind the "
                         "roofline line is computed from")
    ap.add_argument("--host-threads", type=int, default=1, help="host threads enqueueing batches")
    ap.add_argument("--streams-per-thread", type=int, default=3,
                    help="HIP streams each host thread rotates over (batches in flight = threads x this); measured on "
                         "MI355X: 1x3 0.160 ms/step, 1x2 = 2x1 0.173, 3x1 0.167-0.177, 1x4 0.183")
    ap.add_argument("--sample-type", default=None, choices=list(SAMPLE_TYPES),
                    help="default: the workload's (khop2 = the reference's default for GraphSAGE, "
                         "multi_gpu/train_graphsage.py:75)")
    ap.add_argument("--seed", type=lambda s: int(s, 0), default=0x5A4D47)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the HIP path has no CPU fallback")
    local_rank %= max(1, torch.cuda.device_count())  # more ranks than GPUs (a functional check on one GPU) share them
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        # control plane only (two barriers and two 16-byte reductions): the data path has no collective, every rank
        # samples its own step range of the identically shuffled train set.  gloo keeps it off the GPUs entirely.
        import torch.distributed as dist
        dist.init_process_group("gloo")
    lib.load()

    w = WORKLOADS[args.workload]
    if args.sample_type is None:
        args.sample_type = w["sample_type"]
    t_setup = time.time()
    indptr, indices, num_edge = gen_graph_on_gpu(w["num_node"], w["num_edge"], 42, dev)
    feat = gen_features_on_gpu(w["num_node"], w["feat_dim"], dev)
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    label = torch.randint(0, w["num_class"], (w["num_node"],), generator=g, device=dev, dtype=torch.int64)
    train = torch.randperm(w["num_node"], generator=g, device=dev)[:w["num_train"]].to(torch.int32)
    # cache table: top cache_ratio*N nodes by in-degree (stand-in rank list; the split kernel does not care)
    deg = (indptr[1:].to(torch.int64) - indptr[:-1].to(torch.int64)) & 0xFFFFFFFF
    n_cached = int(w["num_node"] * args.cache_ratio)
    table = torch.full((w["num_node"],), -1, dtype=torch.int32, device=dev)
    if n_cached:
        top = torch.argsort(deg, descending=True)[:n_cached]
        table[top] = torch.arange(n_cached, device=dev, dtype=torch.int32)
        del top
    del deg
    # one epoch's shuffle (identical on every rank, like DistShuffler's seed = epoch); each rank takes a step range
    g.manual_seed(0)
    train = train[torch.randperm(train.numel(), generator=g, device=dev)]
    bs = w["batch_size"]
    steps_per_epoch = (train.numel() + bs - 1) // bs
    local_first, _ = local_step_range(steps_per_epoch, rank, world)

    prefix = gen_prefix_on_gpu(indptr, num_edge, 11, dev) if args.sample_type == "weighted_khop_prefix" else None
    prob_t = alias_t = None
    if args.sample_type in ("weighted_khop", "weighted_khop_hash_dedup"):
        prob_t, alias_t = gen_alias_on_gpu(indices, num_edge, 12, dev)
    sampler = lib.Sampler(indptr, indices, w["fanout"], bs, sample_type=SAMPLE_TYPES[args.sample_type], seed=args.seed,
                          prob_prefix=prefix, walk_len=w.get("walk_len", 3), num_walks=w.get("num_walks", 4),
                          restart_prob=w.get("restart_prob", 0.5), prob_table=prob_t, alias_table=alias_t)
    NT = 1 if args.no_overlap else args.host_threads
    SPT = 1 if args.no_overlap else max(1, args.streams_per_thread)
    NBUF = 2 * NT * SPT
    batches = [sampler.new_batch(w["feat_dim"], lib.F32, lib.I64) for _ in range(NBUF)]
    for bt in batches:
        bt.enable_timing(True)  # HIP events around the feature gather, on the stream it is launched on
    # Batches go round-robin over NT x SPT HIP streams (batch i -> stream i % (NT*SPT), enqueued by host thread i % NT;
    # one thread is enough: enqueueing a batch takes ~0.06-0.1 ms): whole
    # batches overlap -- the latency-bound sampling/dedup chain of one with the bandwidth-bound gather of another.
    # fgnn_sampler_run_batch is thread-safe and keeps khop2's in-place CSR swaps in batch order (sequence numbers),
    # so the results are the same as a serial run.  (The reference also overlaps its sample and copy loops.)
    streams = [torch.cuda.Stream(device=dev) for _ in range(NT * SPT)]
    torch.cuda.synchronize()
    t_setup = time.time() - t_setup

    import threading
    metas, gather_ms, host_busy = [], [], [0.0] * NT
    lock = threading.Lock()

    def seeds_of(i):
        step = (local_first + i) % steps_per_epoch
        return step, train[step * bs:min(train.numel(), (step + 1) * bs)]

    extract = [True]  # False: sample + dedup + remap + cache-index split only (the sampler-side stage)

    def worker(t, first, last, timed):
        torch.cuda.set_device(dev)
        if t >= NT:
            return
        mine, gm = [], []
        for i in range(first + ((t - first) % NT), last, NT):
            bt = batches[i % NBUF]
            if i - first >= NBUF:           # buffer reuse: collect the summary of the batch that used it
                m = bt.wait()
                if timed:
                    mine.append(m)
                    gm.append(bt.gather_ms() if extract[0] else -1.0)
            step, seeds = seeds_of(i)
            t_h = time.perf_counter()
            if extract[0]:
                sampler.run_batch(i, seeds, step, bt, table, feat, label, stream=streams[i % len(streams) if NT > 1 or SPT > 1 else 0])
            else:
                sampler.run_batch(i, seeds, step, bt, table, None, None, stream=streams[i % len(streams) if NT > 1 or SPT > 1 else 0])
            host_busy[t] += time.perf_counter() - t_h
        for i in range(max(first, last - NBUF) + ((t - max(first, last - NBUF)) % NT), last, NT):
            m = batches[i % NBUF].wait()
            if timed:
                mine.append(m)
                gm.append(batches[i % NBUF].gather_ms() if extract[0] else -1.0)
        with lock:
            metas.extend(mine)
            gather_ms.extend(gm)

    def run_region(first, last, timed):
        ths = [threading.Thread(target=worker, args=(t, first, last, timed)) for t in range(NT)]
        for th in ths:
            th.start()
        for th in ths:
            th.join()

    def barrier():
        if world > 1:
            torch.distributed.barrier()

    # set-up, not warm-up: a few batches so that code objects are loaded, occupancy queries cached and every buffer
    # touched once even when the caller asks for a very short warm-up (sequence numbers stay consecutive)
    prime = max(0, 12 - args.warmup)
    run_region(0, prime, False)
    torch.cuda.synchronize()
    run_region(prime, prime + args.warmup, False)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    host_busy = [0.0] * NT
    t0 = time.perf_counter()
    run_region(prime + args.warmup, prime + args.warmup + args.steps, True)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    assert len(metas) == args.steps, (len(metas), args.steps)

    # the same kernel with nothing else on the GPU (one thread, one stream): separates the kernel's own efficiency
    # from the slowdown it accepts when it shares the chip with the next batch's sampling chain
    serial = None
    next_seq = prime + args.warmup + args.steps  # sequence numbers must stay consecutive
    metas_t, gather_t = list(metas), list(gather_ms)
    # the sampler-side stage alone (what the reference's kLogEpochSampleTotalTime covers: shuffle slice + sample +
    # dedup + remap + cache-index split, dist_loops_arch5.cc:98-105), same overlap, no feature gather
    metas.clear()
    gather_ms.clear()
    sample_stage = None
    if not args.timed_only:
        extract[0] = False
        n_stage = min(args.steps, 64)
        run_region(next_seq, next_seq + 8, False)
        next_seq += 8
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        run_region(next_seq, next_seq + n_stage, True)
        torch.cuda.synchronize()
        t_stage = time.perf_counter() - t1
        next_seq += n_stage
        stage_edges = sum(int(m.num_edge[l]) for m in metas for l in range(m.num_layers))
        sample_stage = {"edges_per_s": stage_edges / t_stage, "ms_per_step": t_stage / n_stage * 1e3, "steps": n_stage,
                        "note": "sample + dedup + remap + cache-index split only (no feature gather), same overlap"}
        extract[0] = True
    metas.clear()
    gather_ms.clear()
    if (NT > 1 or SPT > 1) and not args.timed_only:
        nt_saved, spt_saved = NT, SPT
        NT = SPT = 1
        base_seq = next_seq
        run_region(base_seq, base_seq + 24, True)
        torch.cuda.synchronize()
        g = [x for x in gather_ms if x >= 0]
        b = sum(int(m.num_input) * (4 + 8 * w["feat_dim"]) for m in metas) / max(len(metas), 1)
        if g:
            ach = b / (float(np.mean(g)) * 1e-3) / 1e9
            serial = {"achieved": ach, "frac": ach / HBM_PEAK_GBS, "avg_launch_ms": float(np.mean(g)), "unit": "GB/s",
                      "note": "same launch with no concurrent batch (1 host thread / stream)"}
        NT, SPT = nt_saved, spt_saved
    metas[:] = metas_t
    gather_ms[:] = gather_t

    # metas hold ctypes structs that alias nothing (copied by value in wait())
    edges = sum(int(m.num_edge[l]) for m in metas for l in range(m.num_layers))
    rows = sum(int(m.num_input) for m in metas)
    overflow = any(m.overflow for m in metas)
    gather_ms = [g for g in gather_ms if g >= 0]
    ab = algorithmic_bytes(metas, w["feat_dim"], bs)
    # dominant kernel = feature gather: U*(4 + 8*D) bytes per launch (index read + row read + row write)
    gather_feat_bytes = sum(int(m.num_input) * (4 + 8 * w["feat_dim"]) for m in metas)
    gather_avg_ms = float(np.mean(gather_ms))
    achieved = gather_feat_bytes / len(metas) / (gather_avg_ms * 1e-3) / 1e9

    elapsed, edges, rows = reduce_over_ranks(elapsed, edges, rows)

    ratio = pmc_traffic_ratio()
    # reference point next to the 8 TB/s spec peak the fraction is quoted against: what torch's plain device-to-device
    # copy of 2 GiB reaches on this GPU right now (read + write bytes per second; ordinary loads/stores -- the gather's
    # non-temporal accesses beat it; tools/gather_sweep.py measured 6.5 TB/s for the gather kernel in isolation)
    copy_gbs = None
    if world == 1:
        a = torch.empty(1 << 29, dtype=torch.float32, device=dev)
        bdst = torch.empty_like(a)
        bdst.copy_(a)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            bdst.copy_(a)
        e1.record()
        torch.cuda.synchronize()
        copy_gbs = 4 * 2 * a.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9
        del a, bdst
    if rank == 0:
        out = {
            "metric": f"sampled-edges/sec ({args.sample_type} fanout {'/'.join(map(str, w['fanout']))}, batch {bs}, full hot "
                      "path: sample + dedup + remap + cache-index split + feature/label gather)",
            "value": edges / elapsed, "unit": "edges/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": f"{args.workload}-shaped synthetic power-law CSR, N={w['num_node']}, "
                                   f"E={num_edge}, feat f32[N,{w['feat_dim']}] resident in HBM, {args.sample_type} fanout "
                                   f"{w['fanout']}, batch {bs}, cache table ratio {args.cache_ratio}, "
                                   f"1 process per GPU, full replica per GPU, disjoint step ranges",
                       "global_batch": bs * world, "parallelism": f"dp{world} (independent samplers)"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": (gather_feat_bytes / len(metas) * ratio) if ratio else None,
                         "traffic_source": "profiles/r01_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, "
                                           "separate passes; per-launch bytes = measured ratio x this run's "
                                           "algorithmic bytes)",
                         "kernel": "gather_rows16_kernel (feature gather)", "avg_launch_ms": gather_avg_ms,
                         "algorithmic_bytes_per_launch": gather_feat_bytes / len(metas),
                         "serial": serial, "torch_copy_GBps": copy_gbs},
            "epoch_time_s": {"sample_plus_extract": steps_per_epoch * (elapsed / args.steps) / world,
                             "note": f"{steps_per_epoch} steps/epoch x ms_per_step / n_gpus; no training step -- the "
                                     "reference's Table 5 'Sample' + 'Extract' columns (0.45 s + 0.35 s on V100s)"},
            "sample_stage": sample_stage if world == 1 else None,
            "rows_per_s": rows / elapsed, "edges_per_step": edges / args.steps / world,
            "input_nodes_per_step": rows / args.steps / world,
            "algorithmic_bytes_per_step": {k: v / len(metas) for k, v in ab.items()},
            "whole_path_hbm_frac": sum(ab.values()) / len(metas) / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS
            if world == 1 else None,
            "overflow": bool(overflow), "setup_s": t_setup,
            "host_threads": NT, "streams": NT * SPT, "host_enqueue_ms_per_step": sum(host_busy) / args.steps * 1e3,
        }
        if world == 1 and not args.no_cpu_baseline:
            if args.sample_type in ("khop2", "khop0"):
                out["cpu_baseline"] = cpu_baseline(w, indptr, indices, feat, train, sample_type=args.sample_type)
            else:
                out["cpu_baseline"] = cpu_baseline_generic(w, args, indptr, indices, prefix, feat, train)
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
