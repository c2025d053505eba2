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
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "fgnn-artifacts_amd"))

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
HOST_LINK_GBS = 64.0    # PCIe Gen5 x16 per direction (secondary bound of the miss rows)
XGMI_LINK_GBS = 153.0   # one xGMI link (sampler -> trainer peer reads)
QUEUE_SLOTS = 170       # messages the shared queue holds at most (mq_size, memory_queue.h:46 = eng_queue.h kMaxSlots)

import torch

from fgnn_hip import lib, synth  # noqa: E402

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


class no_gc:
    """Timed regions run with Python's cyclic garbage collector off (collected right before): a generation-2 pass over
    the process's objects took 40-60 ms when it fell into a 64-batch region (the extract leg read 0.75-0.92 instead of
    0.20 ms per batch for some --steps values and not for others: which allocation crosses the collector's threshold is
    a function of everything allocated before; profiles/r05_i_gc_pause.txt)."""

    def __enter__(self):
        import gc
        gc.collect()
        self.was = gc.isenabled()
        gc.disable()

    def __exit__(self, *exc):
        import gc
        if self.was:
            gc.enable()


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


def gen_powerlaw_on_gpu(num_node, num_edge, seed, device):
    """--graph powerlaw (round 1's generator; no community structure).  Same construction as synth.powerlaw_csr (power-law row lengths, hub-skewed neighbour ids), done with
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


def pmc_traffic():
    """HBM bytes / algorithmic bytes from the newest committed rocprofv3 PMC passes (profiles/r*_pmc_traffic.json,
    written by tools/pmc_summary.py: FETCH_SIZE / WRITE_SIZE corrected as MI355X_MICROARCH.md prescribes, separate
    --pmc passes of this command).  Returns (gather ratio or None, per-stage dict with every kernel family's
    traffic_over_algorithmic or None, file name)."""
    import glob
    names = sorted((os.path.basename(p) for p in glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json"))),
                   reverse=True)  # newest round / tag first
    for name in names:
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                d = json.load(f)
            return d.get("traffic_over_algorithmic"), d.get("per_stage") or d.get("per_kernel"), name
        except Exception:
            continue
    return None, None, None


def pmc_requests(workload):
    """fabric requests per batch of the sampler-side stage from the newest committed counter pass
    (profiles/r*_pmc_requests.json, tools/pmc_requests.sh: TCC_EA0_RDREQ / WRREQ per kernel) for this workload, or None"""
    import glob
    for name in sorted((os.path.basename(p) for p in glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_requests.json"))),
                       reverse=True):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                d = json.load(f)["workloads"].get(workload)
            if d:
                return d, name
        except Exception:
            continue
    return None, None


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


def cpu_baseline(w, indptr, indices, feat, train, budget_s=12.0, sample_type="khop2", cands=None):
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
    elif cands:
        cands = sorted({min(t, os.cpu_count() or 1) for t in cands})
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


def cpu_baseline_products(dev, budget_s=5.0):
    """BASELINE.json config 1 as a recorded number: the reference's arch0 path (CPU sample + CPU extract,
    cpu/cpu_loops.cc:55-227 -- here the reference's own CPU sources in oracle/_ref, never the product) on the
    ogbn-products shape, 2-layer GraphSAGE fanout 10/5 (example/samgraph/train_graphsage.py with --fanout 5 10).  The
    graph is the same R-MAT generator at the products shape, built on `dev` (the GPU when there is one)."""
    from fgnn_hip import rmat
    w = dict(**synth.DATASET_SHAPES["products"], fanout=[10, 5], batch_size=8000)
    t0 = time.time()
    indptr, indices, _ = rmat.rmat_csr(w["num_node"], w["num_edge"], 42, dev)
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    feat = torch.rand((1 << 21, w["feat_dim"]), generator=g, device=dev, dtype=torch.float32)  # masked: 2^21 rows
    train = rmat.train_set(w["num_node"], w["num_train"], 1, dev)
    gen_s = time.time() - t0
    r = cpu_baseline(w, indptr, indices, feat, train, budget_s=budget_s, cands=[16])
    r["config"] = ("BASELINE.json configs[0]: ogbn-products-shaped R-MAT graph (N=%d, E=%d, feat f32[.,%d]), 2-layer "
                   "GraphSAGE fanout 10/5, batch 8000, CPU sample + CPU extract (arch0), graph generated in %.1fs on %s"
                   % (w["num_node"], w["num_edge"], w["feat_dim"], gen_s, dev))
    best = r["all_runs"].get("omp%d" % r["cores"]) or r["all_runs"]["single"]
    steps_per_epoch = (w["num_train"] + w["batch_size"] - 1) // w["batch_size"]
    r["epoch_time_s"] = best["seconds"] / max(best["batches"], 1) * steps_per_epoch  # sample + extract, no training
    return r


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


GRAPH_DESC = {"rmat": "R-MAT (0.57,0.19,0.19,0.05) seed 42, directed, de-duplicated, CSR by destination",
              "powerlaw": "power-law degrees, hub-skewed ids (round 1 generator)"}


def gen_graph_on_gpu(num_node, num_edge, seed, device, graph=None):
    """(indptr, indices, num_edge) of the workload graph on `device`: the R-MAT graph of SURVEY.md 8(d) unless
    graph == "powerlaw" (or FGNN_BENCH_GRAPH=powerlaw)"""
    graph = graph or os.environ.get("FGNN_BENCH_GRAPH", "rmat")
    if graph == "rmat":
        from fgnn_hip import rmat
        return rmat.rmat_csr(num_node, num_edge, seed, device)
    return gen_powerlaw_on_gpu(num_node, num_edge, seed, device)


def gen_graph(args, w, dev):
    """(indptr, indices, num_edge, description) of the workload graph on `dev`"""
    indptr, indices, ne = gen_graph_on_gpu(w["num_node"], w["num_edge"], 42, dev, args.graph)
    return indptr, indices, ne, GRAPH_DESC[args.graph]


def gen_train_set(args, w, dev):
    """train ids (uniform random ids, seed 1), shuffled once like one DistShuffler epoch"""
    from fgnn_hip import rmat
    train = rmat.train_set(w["num_node"], w["num_train"], 1, dev)
    g = torch.Generator(device=dev)
    g.manual_seed(0)
    return train[torch.randperm(train.numel(), generator=g, device=dev)]


# ----------------------------------------------------------------------------------------------------------------------
# N = 1: both halves on one GPU through the kernel-level C ABI

def run_single(args):
    import threading
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    lib.load()
    w = WORKLOADS[args.workload]
    if args.num_walks and "num_walks" in w:
        w = dict(w, num_walks=args.num_walks)
    if args.sample_type is None:
        args.sample_type = w["sample_type"]
    t_setup = time.time()
    indptr, indices, num_edge, graph_desc = gen_graph(args, w, dev)
    feat = gen_features_on_gpu(w["num_node"], w["feat_dim"], dev)
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    label = torch.randint(0, w["num_class"], (w["num_node"],), generator=g, device=dev, dtype=torch.int64)
    train = gen_train_set(args, w, dev)
    bs = w["batch_size"]
    steps_per_epoch = (train.numel() + bs - 1) // bs
    # headline region: cache-index split against a stand-in table (top cache_ratio*N rows by in-degree; the split
    # kernel does not care which rows are cached), every row gathered from the HBM-resident table.  The pre-sampler's
    # table is used by the extract leg below.
    deg = (indptr[1:].to(torch.int64) - indptr[:-1].to(torch.int64)) & 0xFFFFFFFF
    n_cached = int(w["num_node"] * args.cache_ratio)
    table = torch.full((w["num_node"],), -1, dtype=torch.int32, device=dev)
    if n_cached:
        top = torch.argsort(deg, descending=True)[:n_cached]
        table[top] = torch.arange(n_cached, device=dev, dtype=torch.int32)
        del top
    del deg

    prefix = gen_prefix_on_gpu(indptr, num_edge, 11, dev) if args.sample_type == "weighted_khop_prefix" else None
    prob_t = alias_t = None
    if args.sample_type in ("weighted_khop", "weighted_khop_hash_dedup"):
        prob_t, alias_t = gen_alias_on_gpu(indices, num_edge, 12, dev)
    sampler = lib.Sampler(indptr, indices, w["fanout"], bs, sample_type=SAMPLE_TYPES[args.sample_type], seed=args.seed,
                          prob_prefix=prefix, walk_len=w.get("walk_len", 3), num_walks=w.get("num_walks", 4),
                          restart_prob=w.get("restart_prob", 0.5), prob_table=prob_t, alias_table=alias_t)
    NT = 1 if args.no_overlap else args.host_threads
    SPT = 1 if args.no_overlap else max(1, args.streams_per_thread)
    NBUF = max(1, args.buffers_per_stream) * NT * SPT
    batches = [sampler.new_batch(w["feat_dim"], lib.F32, lib.I64) for _ in range(NBUF)]
    # HIP events around the feature gather, on the stream it is launched on -- on every THIRD batch (buffers 0 and 4 of
    # six: streams 0 and 1): the two event records per batch cost the step 3 % when every batch carries them
    # (interleaved A/B, tools/ab_variants.py --timing-variant: 0.1186 -> 0.1222 ms), and the timed region is what
    # `value` is computed from; a third of the launches (~250 per run) is sample enough for their average
    for k, bt in enumerate(batches):
        bt.enable_timing(k % 6 in (0, 4) or len(batches) < 6)
    # Batches go round-robin over NT x SPT HIP streams (batch i -> stream i % (NT*SPT), enqueued by host thread i % NT;
    # one thread is enough: enqueueing a batch takes ~0.06-0.1 ms): whole batches overlap -- the latency-bound
    # sampling/dedup chain of one with the bandwidth-bound gather of another.  fgnn_sampler_run_batch is thread-safe
    # and keeps khop2's in-place CSR swaps in batch order (sequence numbers), so the results are the same as a serial
    # run.  (The reference also overlaps its sample and copy loops.)
    streams = [torch.cuda.Stream(device=dev) for _ in range(NT * SPT)]
    torch.cuda.synchronize()
    t_setup = time.time() - t_setup

    metas, gather_ms, host_busy = [], [], [0.0] * NT
    cached_ms = []
    lock = threading.Lock()

    def seeds_of(i):
        step = i % steps_per_epoch
        return step, train[step * bs:min(train.numel(), (step + 1) * bs)]

    # what a batch does after sampling: "full" = cache split + gather from the HBM table (headline), "sample" = cache
    # split only (sampler-side stage), "cached" = config 3's extract (cache split against the pre-sampler's table,
    # misses from host memory, hits from the HBM cache)
    mode = ["full"]
    leg = {}
    stage_streams = [len(streams)]  # streams the batches rotate over (the sampler-side stage uses fewer, see below)

    def worker(t, first, last, timed):
        torch.cuda.set_device(dev)
        if t >= NT:
            return
        mine, gm, cm = [], [], []

        def collect(bt):
            m = bt.wait()
            if timed:
                mine.append(m)
                gm.append(bt.gather_ms() if mode[0] == "full" else -1.0)
                if mode[0] == "cached":
                    cm.append(bt.extract_cached_ms())
        for i in range(first + ((t - first) % NT), last, NT):
            bt = batches[i % NBUF]
            if i - first >= NBUF:           # buffer reuse: collect the summary of the batch that used it
                collect(bt)
            step, seeds = seeds_of(i)
            st = streams[i % stage_streams[0] if NT > 1 or SPT > 1 else 0]
            t_h = time.perf_counter()
            if mode[0] == "full":
                sampler.run_batch(i, seeds, step, bt, table, feat, label, stream=st)
            elif mode[0] == "sample":
                sampler.run_batch(i, seeds, step, bt, table, None, None, stream=st)
            else:
                sampler.run_batch_cached(i, seeds, step, bt, leg["table"], leg["cache_rows"], leg["host_feat"], label,
                                         stream=st)
            host_busy[t] += time.perf_counter() - t_h
        for i in range(max(first, last - NBUF) + ((t - max(first, last - NBUF)) % NT), last, NT):
            collect(batches[i % NBUF])
        with lock:
            metas.extend(mine)
            gather_ms.extend(gm)
            cached_ms.extend(cm)

    def region_call(first, last):
        """the range as ONE prepared native call (fgnn_sampler_run_range: the reference's loop is a C++ thread too,
        cuda_loops_arch1.cc:38-84) -- no Python, ctypes or GIL work between two batches; .run() is the call itself"""
        sts = streams[:stage_streams[0]] if SPT > 1 else streams[:1]
        if mode[0] == "cached":
            # four batches in flight: a batch's chain here is sampling + split + miss gather (host link, ~0.36 ms) +
            # hit gather, and with three the link idles between miss gathers (0.309 ms per batch, 0.73 of the link;
            # four: 0.294 / 0.77; six: 0.360 -- profiles/r04_f_extract_streams_sweep.txt)
            return sampler.range_call(first, last - first, train, bs, leg["batches"], leg["streams"],
                                      cache_table=leg["table"], label=label, cache_rows=leg["cache_rows"],
                                      full_feat=leg["host_feat"], cached=True)
        return sampler.range_call(first, last - first, train, bs, batches, sts, cache_table=table,
                                  feat=feat if mode[0] == "full" else None,
                                  label=label if mode[0] == "full" else None)

    def absorb(call, timed):
        ms, tm, busy = call.results()
        host_busy[0] += busy
        if timed:
            metas.extend(ms)
            gather_ms.extend(t[0] if mode[0] == "full" else -1.0 for t in tm)
            if mode[0] == "cached":
                cached_ms.extend(tm)

    def run_region(first, last, timed):
        if NT == 1:
            call = region_call(first, last)
            call.run()
            absorb(call, timed)
            return
        ths = [threading.Thread(target=worker, args=(t, first, last, timed)) for t in range(NT)]
        for th in ths:
            th.start()
        for th in ths:
            th.join()

    def timed_region(first, last):
        """seconds for batches first .. last-1, bracketed by a device synchronise on both sides.  One host thread: the
        native call's arguments are marshalled before the clock starts and its per-batch summaries are turned into
        Python objects after it stops -- the bracket holds the native loop over the batches and the synchronise,
        nothing else (the wrapper's Python around the call measured ~0.15 ms: 6 % of a 20-batch window)."""
        call = region_call(first, last) if NT == 1 else None
        torch.cuda.synchronize()
        with no_gc():
            t0 = time.perf_counter()
            if call is not None:
                call.run()
            else:
                run_region(first, last, True)
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
        if call is not None:
            absorb(call, True)
        return el

    # set-up, not warm-up: a few batches so that code objects are loaded, occupancy queries cached and every buffer
    # touched once even when the caller asks for a very short warm-up (sequence numbers stay consecutive)
    prime = max(0, 12 - args.warmup)
    run_region(0, prime, False)
    torch.cuda.synchronize()
    run_region(prime, prime + args.warmup, False)
    torch.cuda.synchronize()
    # what the memory system sustains for the sampling chain's access pattern HERE and NOW: independent random 4-byte
    # reads from the CSR (6.5 GB: far beyond every cache), ~20 ms, nothing else on the GPU (roofline_sample's ceiling)
    probe_reads_per_s = None
    try:
        probe_reads_per_s = lib.random_read_rate(indices)
    except Exception:
        pass
    # R timed windows of EXACTLY args.steps steps each, back to back (a 151-step window is ~20 ms: one window is a thin
    # basis for a headline); every window is bracketed by a device synchronise on both sides, all R values are
    # published and `value` is the MEDIAN window's
    R = max(1, args.windows)
    windows = []
    seq0 = prime + args.warmup
    for r in range(R):
        metas.clear()
        gather_ms.clear()
        for t in range(NT):
            host_busy[t] = 0.0
        el = timed_region(seq0 + r * args.steps, seq0 + (r + 1) * args.steps)
        assert len(metas) == args.steps, (len(metas), args.steps)
        windows.append(dict(elapsed=el, metas=list(metas), gather_ms=list(gather_ms),
                            host_enqueue_ms=sum(host_busy) / args.steps * 1e3))
    order = sorted(range(R), key=lambda r: windows[r]["elapsed"])
    med = windows[order[(R - 1) // 2]]  # the median window (the slower of the middle two for an even R)
    elapsed = med["elapsed"]
    host_enqueue_ms = med["host_enqueue_ms"]  # of that timed window only
    metas[:] = med["metas"]
    gather_ms[:] = med["gather_ms"]
    window_ms = [wd["elapsed"] / args.steps * 1e3 for wd in windows]
    del windows

    next_seq = prime + args.warmup + R * args.steps  # sequence numbers must stay consecutive
    metas_t, gather_t = list(metas), list(gather_ms)
    # the sampler-side stage alone (what the reference's kLogEpochSampleTotalTime covers: shuffle slice + sample +
    # dedup + remap + cache-index split, dist_loops_arch5.cc:98-105), same overlap, no feature gather
    metas.clear()
    gather_ms.clear()
    sample_stage = None
    if not args.timed_only:
        mode[0] = "sample"
        # two batch streams: without the gather the stage is bound by khop2's order chain and a third batch in flight only
        # slows the chain's kernels (0.071 ms per batch against 0.076 with three; an arch5 sampler process, which also
        # packs and publishes every batch, does better with three: profiles/r05_m_sampler_streams_sweep.txt)
        if NT == 1 and 0 < args.stage_streams < SPT:
            stage_streams[0] = args.stage_streams
        n_stage = min(args.steps, 64)
        run_region(next_seq, next_seq + 8, False)
        next_seq += 8
        t_stage = timed_region(next_seq, next_seq + n_stage)
        next_seq += n_stage
        stage_edges = sum(int(m.num_edge[l]) for m in metas for l in range(m.num_layers))
        ab_s = algorithmic_bytes(metas, w["feat_dim"], bs)
        stage_bytes = (ab_s["sample"] + ab_s["dedup_remap"] + ab_s["cache_split"]) / max(len(metas), 1)
        sample_stage = {"edges_per_s": stage_edges / t_stage, "ms_per_step": t_stage / n_stage * 1e3, "steps": n_stage,
                        "algorithmic_bytes_per_step": stage_bytes,
                        "hbm_frac": stage_bytes / (t_stage / n_stage) / 1e9 / HBM_PEAK_GBS,
                        "streams": stage_streams[0],
                        "note": "sample + dedup + remap + cache-index split only (no feature gather), batches over "
                                "%d streams (the stage's optimum is two; an arch5 sampler process, which also packs "
                                "and publishes, uses three)" % stage_streams[0]}
        mode[0] = "full"
        stage_streams[0] = len(streams)
    # the latency-bound stage against the chip's random-request rate: fabric requests per batch (committed counter pass
    # of this workload) / the stage's time, against what the probe above sustained in this very run
    roofline_sample = None
    req, req_file = pmc_requests(args.workload) if args.sample_type == w["sample_type"] and args.graph == "rmat" else (None, None)
    if sample_stage and req and probe_reads_per_s:
        per_read = req.get("probe_requests_per_read") or 1.0
        side = req["sampler_side_per_batch"]
        total_req = side["read"] + side["write"]
        ach = total_req / (sample_stage["ms_per_step"] * 1e-3)
        peak = probe_reads_per_s * per_read
        worst = sorted(((k, v["read_per_batch"] + v["write_per_batch"]) for k, v in req["kernels"].items()
                        if not k.startswith("gather_rows")), key=lambda kv: -kv[1])
        roofline_sample = {
            "bound": "fabric random-request rate", "requests_per_batch": total_req, "read_requests_per_batch": side["read"],
            "write_requests_per_batch": side["write"], "achieved": ach / 1e9, "peak": peak / 1e9, "unit": "G requests/s",
            "frac": ach / peak, "stage_ms_per_step": sample_stage["ms_per_step"],
            "probe": {"random_reads_per_s": probe_reads_per_s, "requests_per_read": per_read,
                      "what": "fgnn_debug_random_reads: independent random 4-byte reads from this run's CSR array, four "
                              "in flight per lane, alone on the GPU, in this run's warm-up"},
            "requests_by_kernel_per_batch": {k: v for k, v in worst},
            "requests_source": "profiles/%s (rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum, tools/pmc_requests.sh; "
                               "counts of the one-stream run)" % req_file,
            "note": "sampler-side stage = sample + dedup + remap + cache-index split; the stage is a chain of dependent "
                    "random accesses, so the chip's random-request rate, not HBM bytes, is what bounds it"}
    metas.clear()
    gather_ms.clear()
    # the gather with nothing else on the GPU (one thread, one stream): separates the kernel's own efficiency from
    # the slowdown it accepts when it shares the chip with the next batch's sampling chain
    serial = None
    if (NT > 1 or SPT > 1) and not args.timed_only:
        nt_saved, spt_saved = NT, SPT
        NT = SPT = 1
        with no_gc():
            run_region(next_seq, next_seq + 24, True)
        next_seq += 24
        torch.cuda.synchronize()
        gsel = [x for x in gather_ms if x >= 0]
        b = sum(int(m.num_input) * (4 + 8 * w["feat_dim"]) for m in metas) / max(len(metas), 1)
        if gsel:
            ach = b / (float(np.mean(gsel)) * 1e-3) / 1e9
            serial = {"achieved": ach, "frac": ach / HBM_PEAK_GBS, "avg_launch_ms": float(np.mean(gsel)), "unit": "GB/s",
                      "note": "same launch with no concurrent batch (1 host thread / stream)"}
        NT, SPT = nt_saved, spt_saved
    metas.clear()
    gather_ms.clear()

    # ---- BASELINE config 3's extract leg on this GPU: features in HOST memory, HBM cache of the top cache_ratio*N
    # rows ranked by the pre-sampler (dist/pre_sampler.cc:75-162 -> fgnn_presample_count / fgnn_presample_rank), hit
    # rows from the cache, miss rows read by the gather kernel over the host link (dist_loops.cc:713-846)
    extract_leg = None
    if args.cache_ratio > 0 and not args.timed_only and not args.no_extract_leg and args.sample_type in ("khop2", "khop0"):
        try:
            extract_leg, next_seq = run_extract_leg(args, w, dev, sampler, batches, streams, train, feat, label,
                                                    steps_per_epoch, next_seq, run_region, timed_region, mode, leg, metas,
                                                    cached_ms)
        except Exception as e:  # the headline must not be lost to a problem in a secondary measurement
            extract_leg = {"error": "%s: %s" % (type(e).__name__, e)}
    # ---- the epoch WITH training on this one GPU (config 2's shape: one MI355X samples, extracts and trains): the
    # next batch's sample + extract chain runs on a side stream under the current batch's GraphSAGE step
    # (examples/models.py, hidden 256, fused Adam), like the reference's arch3 threads
    train_leg = None
    if not args.timed_only and not args.no_train_leg and args.sample_type != "random_walk":
        try:
            train_leg, next_seq = run_train_leg(args, w, dev, sampler, batches, streams, seeds_of, table, feat, label,
                                                steps_per_epoch, next_seq, mode)
        except Exception as e:
            train_leg = {"error": "%s: %s" % (type(e).__name__, e)}
    metas[:] = metas_t
    gather_ms[:] = gather_t

    # metas hold ctypes structs that alias nothing (copied by value in wait())
    edges = sum(int(m.num_edge[l]) for m in metas for l in range(m.num_layers))
    rows = sum(int(m.num_input) for m in metas)
    overflow = any(m.overflow for m in metas)
    gather_ms = [x for x in gather_ms if x >= 0]
    ab = algorithmic_bytes(metas, w["feat_dim"], bs)
    # dominant kernel = feature gather: U*(4 + 8*D) bytes per launch (index read + row read + row write)
    gather_feat_bytes = sum(int(m.num_input) * (4 + 8 * w["feat_dim"]) for m in metas)
    gather_avg_ms = float(np.mean(gather_ms))
    achieved = gather_feat_bytes / len(metas) / (gather_avg_ms * 1e-3) / 1e9

    ratio, per_kernel, pmc_file = pmc_traffic()
    if (args.workload, args.sample_type, args.graph) != ("papers100M", "khop2", "rmat"):
        # the PMC passes were taken on the default workload: the gather's ratio (a property of the kernel: rows are
        # whole cache lines) carries over, the sampler-side per-stage ratios do not
        per_kernel = None
    # reference point next to the 8 TB/s spec peak the fraction is quoted against: what torch's plain device-to-device
    # copy of 2 GiB reaches on this GPU right now (read + write bytes per second; ordinary loads/stores -- the gather's
    # non-temporal accesses beat it)
    a = torch.empty(1 << 29, dtype=torch.float32, device=dev)
    bdst = torch.empty_like(a)
    bdst.copy_(a)
    copies = []
    for _ in range(5):  # five measurements of 4 copies each: the spread tells a noisy box from a slow one
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            bdst.copy_(a)
        e1.record()
        torch.cuda.synchronize()
        copies.append(4 * 2 * a.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9)
    copy_gbs = float(np.median(copies))
    del a, bdst
    out = {
        "metric": f"sampled-edges/sec ({args.sample_type} fanout {'/'.join(map(str, w['fanout']))}, batch {bs}, full hot "
                  f"path: sample + dedup + remap + cache-index split + feature/label gather; median of {R} timed windows "
                  f"of {args.steps} steps)",
        "value": edges / elapsed, "unit": "edges/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
        "windows": {"count": R, "ms_per_step": window_ms, "min": min(window_ms), "max": max(window_ms),
                    "note": "every window: args.steps steps between two device synchronisations; value / ms_per_step / "
                            "roofline come from the median window"},
        "vs_baseline": None, "dtype": "u32", "data": "synthetic",
        "config": {"workload": f"{args.workload}-shaped synthetic graph: {graph_desc}; N={w['num_node']}, "
                               f"E={num_edge}, train set {w['num_train']} uniform random ids (seed 1), feat "
                               f"f32[N,{w['feat_dim']}] resident in HBM, {args.sample_type} fanout {w['fanout']}"
                               + (f" ({w['num_walks']} walks x {w['walk_len']} steps, restart {w['restart_prob']})"
                                  if args.sample_type == "random_walk" else "") + ", batch "
                               f"{bs}, cache table ratio {args.cache_ratio}, 1 GPU samples and extracts",
                   "global_batch": bs, "parallelism": "1 GPU (sampler + extractor)"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS,
                     "traffic": (gather_feat_bytes / len(metas) * ratio) if ratio else None,
                     "traffic_source": f"profiles/{pmc_file} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; "
                                       "per-launch bytes = measured ratio x this run's algorithmic bytes)"
                     if pmc_file else None,
                     "traffic_over_algorithmic_per_kernel": per_kernel,
                     "kernel": "gather_rows16_kernel (feature gather)", "avg_launch_ms": gather_avg_ms,
                     "timed_launches": len(gather_ms),
                     "timing": "HIP events on the launch's own stream around every third batch's gather inside the timed "
                               "window (event records on every batch cost the step 3 %)",
                     "algorithmic_bytes_per_launch": gather_feat_bytes / len(metas),
                     "serial": serial, "torch_copy_GBps": copy_gbs,
                     "torch_copy_GBps_spread": {"min": min(copies), "max": max(copies), "samples": copies}},
        "roofline_extract": extract_leg,
        # the N >= 2 lines measure the factored pipeline with the features in HOST memory behind a cache_ratio cache; the
        # same work on ONE GPU (this process samples AND does the cached extraction with host misses) is the N = 1 point
        # of that curve -- `value` above is config 2's shape (features HBM-resident) and is not comparable with N >= 2
        "pipeline_n1_point": ({"value": (edges / args.steps) / (extract_leg["ms_per_step"] * 1e-3), "unit": "edges/s",
                               "ms_per_step": extract_leg["ms_per_step"],
                               "what": "sample + dedup + remap + cache split + cached extraction (hits from the HBM cache, "
                                       "misses over the host link) on one GPU: the like-for-like N = 1 point of the "
                                       "--gpus N >= 2 pipeline lines"}
                              if extract_leg and "ms_per_step" in extract_leg else None),
        "epoch_time_s": {"sample_plus_extract": steps_per_epoch * (elapsed / args.steps),
                         "sample_plus_extract_cache_0.2_host_misses":
                             steps_per_epoch * extract_leg["ms_per_step"] * 1e-3
                             if extract_leg and "ms_per_step" in extract_leg else None,
                         "with_training": steps_per_epoch * train_leg["ms_per_step"] * 1e-3
                             if train_leg and "ms_per_step" in train_leg else None,
                         "note": f"{steps_per_epoch} steps/epoch x ms_per_step; sample_plus_extract = the reference's "
                                 "Table 5 'Sample' + 'Extract' columns (0.45 s + 0.35 s on V100s); with_training = the "
                                 "same batches with a GraphSAGE step each on this GPU (train_leg; the reference: 0.28 s "
                                 "on 8 V100s, exp/table4); the factored pipeline's epoch is measured by the N >= 2 runs"},
        "train_leg": train_leg,
        "sample_stage": sample_stage,
        "roofline_sample": roofline_sample,
        "probe_random_reads_per_s": probe_reads_per_s,
        "rows_per_s": rows / elapsed, "edges_per_step": edges / args.steps,
        "input_nodes_per_step": rows / args.steps,
        "algorithmic_bytes_per_step": {k: v / len(metas) for k, v in ab.items()},
        "whole_path_hbm_frac": sum(ab.values()) / len(metas) / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS,
        "overflow": bool(overflow), "setup_s": t_setup,
        "host_threads": NT, "streams": NT * SPT, "host_enqueue_ms_per_step": host_enqueue_ms,
    }
    if not args.no_cpu_baseline:
        if args.sample_type in ("khop2", "khop0"):
            out["cpu_baseline"] = cpu_baseline(w, indptr, indices, feat, train, sample_type=args.sample_type)
        else:
            out["cpu_baseline"] = cpu_baseline_generic(w, args, indptr, indices, prefix, feat, train)
        if args.workload == "papers100M":
            # every BASELINE.json config has a recorded line: configs[0] is the reference's CPU-runnable case
            del indptr, indices, feat, label, table, sampler, batches
            torch.cuda.empty_cache()
            try:
                out["cpu_baseline_products"] = cpu_baseline_products(dev)
            except Exception as e:
                out["cpu_baseline_products"] = {"error": "%s: %s" % (type(e).__name__, e)}
    # the line is long and a log tail shows its END: the figures a reader looks for first, once more, last
    out["summary"] = {"value": out["value"], "unit": out["unit"], "ms_per_step": out["ms_per_step"],
                      "windows_ms_per_step": [round(x, 5) for x in window_ms], "steps": args.steps, "warmup": args.warmup,
                      "host_enqueue_ms_per_step": host_enqueue_ms, "gather_frac_of_hbm_peak": out["roofline"]["frac"],
                      "sample_stage_ms_per_step": sample_stage["ms_per_step"] if sample_stage else None,
                      "extract_leg_ms_per_step": extract_leg.get("ms_per_step") if extract_leg else None,
                      "train_leg_ms_per_step": train_leg.get("ms_per_step") if train_leg else None,
                      "cpu_baseline_edges_per_s": (out.get("cpu_baseline") or {}).get("value")}
    print(json.dumps(out), flush=True)


# ---- NUMA placement of host tables the GPU reads over the host link ---------------------------------------------------
def gpu_numa_node(dev_id):
    """NUMA node the GPU's PCIe root hangs off (sysfs), or None"""
    try:
        p = torch.cuda.get_device_properties(dev_id)
        bdf = "%04x:%02x:%02x.0" % (getattr(p, "pci_domain_id", 0), p.pci_bus_id, p.pci_device_id)
        with open("/sys/bus/pci/devices/%s/numa_node" % bdf) as f:
            n = int(f.read().strip())
        return n if n >= 0 else None
    except Exception:
        return None


def numa_nodes_with_memory():
    try:
        txt = open("/sys/devices/system/node/has_memory").read().strip()
        out = []
        for part in txt.split(","):
            a, _, b = part.partition("-")
            out += list(range(int(a), int(b or a) + 1))
        return out
    except Exception:
        return []


def pages_by_numa_node(addr, nbytes, samples=1024):
    """{node: pages} over `samples` evenly spaced pages of [addr, addr + nbytes) (move_pages in query mode)"""
    import ctypes as C
    try:
        numa = C.CDLL("libnuma.so.1")
        n = max(1, min(samples, nbytes // 4096))
        base = addr & ~4095
        pages = (C.c_void_p * n)(*[base + (i * (nbytes // n) & ~4095) for i in range(n)])
        status = (C.c_int * n)()
        numa.numa_move_pages.argtypes = [C.c_int, C.c_ulong, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                         C.c_int]
        if numa.numa_move_pages(0, n, pages, None, status, 0) != 0:
            return None
        hist = {}
        for v in status:
            hist[int(v)] = hist.get(int(v), 0) + 1
        return {str(k): v for k, v in sorted(hist.items())}
    except Exception:
        return None


class HostTable:
    """A pinned, GPU-readable host array placed on a chosen NUMA node: numa_alloc_onnode + first touch + hipHostRegister.
    torch's pin_memory (hipHostMalloc) leaves the placement to the runtime; on a two-socket host a table on the far
    socket costs the GPU's row reads the inter-socket hop (BENCH_r02: 35.9 GB/s on one box, 49-55 on others)."""

    def __init__(self, rows, dim, node):
        import ctypes as C
        self.C, self.nbytes, self.node = C, rows * dim * 4, node
        self.numa = C.CDLL("libnuma.so.1")
        self.numa.numa_alloc_onnode.restype = C.c_void_p
        self.numa.numa_alloc_onnode.argtypes = [C.c_size_t, C.c_int]
        self.numa.numa_free.argtypes = [C.c_void_p, C.c_size_t]
        self.ptr = self.numa.numa_alloc_onnode(self.nbytes, node)
        if not self.ptr:
            raise MemoryError("numa_alloc_onnode(%d bytes, node %d)" % (self.nbytes, node))
        self.array = np.frombuffer((C.c_char * self.nbytes).from_address(self.ptr), dtype=np.float32).reshape(rows, dim)
        self.array[:] = 0  # first touch under the node binding
        self.hip = C.CDLL("libamdhip64.so")
        self.registered = False
        rc = self.hip.hipHostRegister(C.c_void_p(self.ptr), C.c_size_t(self.nbytes), C.c_uint(3))  # portable | mapped
        if rc != 0:
            self.free()
            raise RuntimeError("hipHostRegister failed with %d" % rc)
        self.registered = True
        d = C.c_void_p()
        rc = self.hip.hipHostGetDevicePointer(C.byref(d), C.c_void_p(self.ptr), C.c_uint(0))
        if rc != 0 or not d.value:
            self.free()
            raise RuntimeError("hipHostGetDevicePointer failed with %d" % rc)
        self.device_ptr = d.value
        self.tensor = torch.from_numpy(self.array)

    def free(self):
        if self.registered:
            self.hip.hipHostUnregister(self.C.c_void_p(self.ptr))
            self.registered = False
        if self.ptr:
            self.tensor = self.array = None
            self.numa.numa_free(self.C.c_void_p(self.ptr), self.nbytes)
            self.ptr = None


def run_extract_leg(args, w, dev, sampler, batches, streams, train, feat, label, steps_per_epoch, next_seq, run_region,
                    timed_region, mode, leg, metas, cached_ms):
    """BASELINE config 3's trainer-side leg on this GPU.  PRIMARY = the contract configuration, presample_epoch =
    args.presample_epochs (default 1: the reference's default, common_config.py:70, and SURVEY 8(d)); every value of
    --presample-variants (default 3) is measured the same way afterwards and reported under `variants` -- a longer
    ranking is a CONFIGURATION change (higher hit rate, fewer host-link bytes), not a kernel change."""
    bs = w["batch_size"]
    num_node, dim = w["num_node"], w["feat_dim"]
    t_init = time.time()
    freq = torch.zeros(num_node, dtype=torch.int32, device=dev)
    bt = batches[0]
    n_cached = int(num_node * args.cache_ratio)
    # host feature table: 2^k rows, node ids masked (SAMGRAPH_EMPTY_FEAT / the reference's papers100M_empty): the full
    # 57 GB table is not needed to exercise random host-DRAM row reads; 2^24 rows x 512 B = 8.6 GB is far beyond any cache
    bits = min(args.empty_feat_bits, int(np.floor(np.log2(num_node))))
    mask = (1 << bits) - 1
    # on the GPU's NUMA node when the host has several (--host-feat-numa auto): one consumer, so next to it
    gnode, nodes = gpu_numa_node(dev.index or 0), numa_nodes_with_memory()
    want = args.host_feat_numa
    table_obj, placement = None, "torch pin_memory (hipHostMalloc; placement left to the runtime)"
    node = gnode if want in ("auto", "gpu") else int(want[5:]) if want.startswith("node:") else None
    if node is not None and len(nodes) > 1 and node in nodes:
        try:
            table_obj = HostTable(1 << bits, dim, node)
            host_feat = table_obj.tensor
            placement = "numa_alloc_onnode(node %d) + first touch + hipHostRegister" % node
        except Exception as e:
            placement += "; node-local allocation failed: %s" % e
            table_obj = None
    if table_obj is None:
        host_feat = torch.empty((1 << bits, dim), dtype=torch.float32).pin_memory()
    host_feat.copy_(feat[:1 << bits])
    numa_info = {"gpu_node": gnode, "nodes_with_memory": nodes, "host_feat_placement": placement,
                 "host_feat_pages_by_node": pages_by_numa_node(host_feat.data_ptr(), host_feat.numel() * 4),
                 "policy_requested": want}
    n_leg_streams = 4 if len(streams) == 3 else len(streams)
    leg_streams = list(streams) + [torch.cuda.Stream(device=dev) for _ in range(n_leg_streams - len(streams))]
    leg_batches = list(batches) + [sampler.new_batch(dim, lib.F32, lib.I64)
                                   for _ in range(max(0, 2 * n_leg_streams - len(batches)))]
    for k, b in enumerate(leg_batches[len(batches):]):
        b.enable_timing(k == 0)
    for b in leg_batches:
        lib.load().fgnn_batch_set_feat_row_mask(b.h, mask)
    cache_rows = torch.empty((n_cached, dim), dtype=torch.float32, device=dev)
    row_b = dim * 4
    state = {"epochs": 0, "seq": next_seq, "presample_s": 0.0}

    def presample_to(epochs):
        """continue the pre-sampling up to `epochs` epochs (keys of their own so that the draws differ from the measured
        batches', eng_engine.cc:PreSample; RunConfig::presample_epoch), rank, rebuild the cache"""
        t0 = time.time()
        with torch.cuda.stream(streams[0]):
            for step in range(steps_per_epoch * state["epochs"], steps_per_epoch * epochs):
                s0 = step % steps_per_epoch
                seeds = train[s0 * bs:min(train.numel(), (s0 + 1) * bs)]
                sampler.sample(seeds, (1 << 63) | step, bt, seq=state["seq"])
                state["seq"] += 1
                lib.presample_count(freq, bt.input_nodes_buffer(), d_num_nodes=bt.d_num_input())
            bt.finish()
            bt.wait()
            rank = lib.presample_rank(freq)
            ptable = lib.cache_table_build(rank, n_cached)
            # the cache holds the rows the trainer would read for the cached nodes: feat[rank[i] & mask]
            lib.gather_rows(cache_rows, feat, src_index=rank[:n_cached], src_row_mask=mask)
            streams[0].synchronize()
        state["epochs"] = epochs
        state["presample_s"] += time.time() - t0
        return rank, ptable

    def measure(epochs, ptable, checked):
        # what the kernels get: the device-visible address of the table (== the host address for hipHostMalloc memory)
        leg.update(table=ptable, cache_rows=cache_rows, streams=leg_streams, batches=leg_batches,
                   host_feat=lib.DevicePointer(table_obj.device_ptr, host_feat) if table_obj else host_feat)
        torch.cuda.synchronize()
        mode[0] = "cached"
        if not checked:
            # correctness of the leg, once per ranking: every row of one batch equals feat[input_nodes & mask]
            step0 = 3
            sampler.run_batch_cached(state["seq"], train[step0 * bs:(step0 + 1) * bs], step0, bt, ptable, cache_rows,
                                     leg["host_feat"], label, stream=streams[0])
            state["seq"] += 1
            bt.wait()
            torch.cuda.synchronize()
            ref = feat[(bt.input_nodes().to(torch.int64) & 0xFFFFFFFF) & mask]
            if not torch.equal(bt.feat(), ref):
                raise RuntimeError("cached extraction differs from the direct gather")
            del ref
        n = min(args.steps, 64)
        metas.clear()
        cached_ms.clear()
        run_region(state["seq"], state["seq"] + 8, False)
        state["seq"] += 8
        dt = timed_region(state["seq"], state["seq"] + n)
        state["seq"] += n
        edges = sum(int(m.num_edge[l]) for m in metas for l in range(m.num_layers))
        rows = sum(int(m.num_input) for m in metas)
        miss = sum(int(m.num_miss) for m in metas)
        hit = sum(int(m.num_cache) for m in metas)
        ms_miss = [a for a, _ in cached_ms if a >= 0]
        ms_hit = [b for _, b in cached_ms if b >= 0]
        hit_bytes = hit * (2 * row_b + 8)
        return {
            "presample_epoch": epochs,
            "workload": f"features in host memory ({1 << bits} rows, ids masked), HBM cache of {n_cached} rows "
                        f"(ratio {args.cache_ratio}) ranked by the pre-sampler over {epochs} epoch(s), same batches as "
                        f"the headline, {n_leg_streams} batches in flight",
            "streams": n_leg_streams,
            "steps": n, "ms_per_step": dt / n * 1e3, "edges_per_s": edges / dt, "rows_per_s": rows / dt,
            "hit_rate": hit / max(rows, 1), "miss_rows_per_step": miss / n, "hit_rows_per_step": hit / n,
            "kernel": "extract_fused_kernel: ONE launch per batch -- a band of %d workgroups pulls the miss rows over "
                      "the host link while the rest of the grid streams the hit rows from the HBM cache; labels and the "
                      "batch summary ride in the HBM band (SURVEY 8(f) rank 1)" % lib.LINK_WGS_SHARED,
            "miss": {"bound": "host link", "bytes_per_step": miss * row_b / n,
                     "achieved": miss * row_b / dt / 1e9, "peak": HOST_LINK_GBS, "unit": "GB/s",
                     "frac": miss * row_b / dt / 1e9 / HOST_LINK_GBS,
                     "band_ms": float(np.mean(ms_miss)) if ms_miss else None,
                     "note": "host-link bytes = miss rows x row bytes over the WALL time of the region (all batches; "
                             "this GPU also samples them); band_ms = first start .. last end of the link band's "
                             "workgroups inside one launch (device clock; bands of up to %d batches share the link)"
                             % n_leg_streams},
            "cached": {"bound": "hbm", "bytes_per_step": hit_bytes / n,
                       "band_ms": float(np.mean(ms_hit)) if ms_hit else None,
                       "achieved": (hit_bytes / n) / (float(np.mean(ms_hit)) * 1e-3) / 1e9 if ms_hit else None,
                       "peak": HBM_PEAK_GBS, "unit": "GB/s",
                       "frac": (hit_bytes / n) / (float(np.mean(ms_hit)) * 1e-3) / 1e9 / HBM_PEAK_GBS if ms_hit else None,
                       "note": "HBM band of the launch: hit rows x (read + write + 2 index words) / the band's first "
                               "start .. last end (device clock stamps of its workgroups, timed batches only)"},
        }

    primary_epochs = max(1, args.presample_epochs)
    variants = sorted({int(v) for v in str(args.presample_variants).split(",") if v.strip()} - {primary_epochs})
    results = {}
    rank1 = None
    for ep in sorted({primary_epochs, *variants}):
        rank, ptable = presample_to(ep)
        results[ep] = measure(ep, ptable, checked=False)
        if ep == primary_epochs:
            rank1 = rank
        else:
            del rank
        del ptable
    res = results[primary_epochs]
    res.update({"presample_s": state["presample_s"], "init_s": time.time() - t_init,
                "checked": "one batch per ranking compared row by row with the direct gather", "numa": numa_info,
                "contract": "presample_epoch = %d%s" % (primary_epochs, " (SURVEY 8(d); reference default, "
                            "example/samgraph/common_config.py:70)" if primary_epochs == 1 else " (NOT the contract's 1)"),
                "variants": {"presample_epoch_%d" % ep: {k: results[ep][k] for k in
                                                         ("ms_per_step", "hit_rate", "miss_rows_per_step", "edges_per_s",
                                                          "miss", "cached")}
                             for ep in variants},
                "variants_note": "same kernels, same batches, a longer pre-sampling ranking (the reference's runner "
                                 "sweeps 1-3, exp/common/runner_helper.py:47-49): a configuration change"})
    next_seq = state["seq"]
    rank = rank1
    # How good is the pre-sampler's ranking?  One more epoch of sampling, counted: the hit rate of the pre-sampler's
    # cache on THAT epoch next to the cache that knows the epoch in advance (the reference's cache-by-fake-optimal tool,
    # utility/data-process/toolkit/cache/cache_by_fake_optimal.cc:66-185: rank by the frequencies of the measured
    # epochs themselves), all at the same ratio
    try:
        freq2 = torch.zeros(num_node, dtype=torch.int32, device=dev)
        with torch.cuda.stream(streams[0]):
            for step in range(steps_per_epoch):
                seeds = train[step * bs:min(train.numel(), (step + 1) * bs)]
                sampler.sample(seeds, (1 << 62) | step, bt, seq=next_seq)
                next_seq += 1
                lib.presample_count(freq2, bt.input_nodes_buffer(), d_num_nodes=bt.d_num_input())
            bt.finish()
            bt.wait()
            total = float(freq2.sum(dtype=torch.int64))
            f64 = freq2.to(torch.int64)
            hit_pre = float(f64[(rank[:n_cached].to(torch.int64) & 0xFFFFFFFF)].sum()) / total
            rank2 = lib.presample_rank(freq2)
            hit_opt = float(f64[(rank2[:n_cached].to(torch.int64) & 0xFFFFFFFF)].sum()) / total
            streams[0].synchronize()
        res["hit_rate_by_policy"] = {
            "pre_sample (%d epoch(s), what the leg above used)" % primary_epochs: hit_pre,
            "fake_optimal (hindsight on the same epoch)": hit_opt,
            "note": "row-weighted hit rates of one further sampled epoch at cache ratio %.2f; fake_optimal ranks by that "
                    "epoch's own frequencies (cache_by_fake_optimal.cc), an upper bound for any static cache" % args.cache_ratio}
        del freq2, f64, rank2
    except Exception as e:  # a secondary figure must not cost the leg
        res["hit_rate_by_policy"] = {"error": "%s: %s" % (type(e).__name__, e)}
    for b in batches:
        lib.load().fgnn_batch_set_feat_row_mask(b.h, 0xFFFFFFFF)
    mode[0] = "full"
    leg.clear()
    del host_feat, cache_rows, freq, rank, rank1
    if table_obj is not None:
        torch.cuda.synchronize()
        table_obj.free()
    return res, next_seq


# ----------------------------------------------------------------------------------------------------------------------
# N >= 2: the factored pipeline, one process per GPU

def run_train_leg(args, w, dev, sampler, batches, streams, seeds_of, table, feat, label, steps_per_epoch, next_seq, mode):
    """K2 batches: sample + extract (all features in HBM) on a side stream, one batch ahead of a GraphSAGE training step
    on torch's current stream.  Returns ({ms_per_step, ...}, next sequence number)."""
    sys.path.insert(0, os.path.join(ROOT, "examples"))
    from models import MODELS
    from samgraph.torch.adapter import CooBlock
    L = len(w["fanout"])
    from graphed_step import GraphedSageStep
    model = MODELS["graphsage"](w["feat_dim"], 256, w["num_class"], L, 0.5).to(dev)
    loss_fcn = torch.nn.CrossEntropyLoss()
    graphed = not args.eager_train
    # Adam as ONE launch with the step count on the device (fgnn_hip.nn.Adam = torch.optim.Adam's update, checked
    # against it in tests/test_train_ops_gpu.py); its step count also keys the fused ReLU + dropout masks
    from fgnn_hip.nn import Adam as FusedAdam
    opt = FusedAdam(model.parameters(), lr=0.003)
    model.dropout_step = opt.step_count
    model.train()
    # the step replayed as a captured HIP graph (examples/graphed_step.py); --eager-train: op by op like the reference's
    # loop.  Eager, the step is bound by the ~40 ops Python launches (0.83 ms of host time for ~0.5 ms of kernels); a
    # replayed graph costs the host 0.12 ms and the GPU 0.58 ms (round 4; 28 nodes and ~0.33 ms since round 5's one-launch
    # pieces and GEMM choice).  (Every graph node costs the GPU 15-20 us on this
    # runtime: with the 45 nodes of the op-by-op SAGEConv layers the replay took 0.93 ms and lost to eager; the fused
    # layer of examples/models.py is what made the graph worth it -- profiles/r04_c_train_graph_vs_eager.txt.)
    stepper = (GraphedSageStep(model, opt, loss_fcn, w["batch_size"], tune_gemms=not args.no_gemm_tuning)
               if graphed else None)
    mode[0] = "full"
    warm, timed = train_region_batches(args.steps, args.train_steps, 1)
    warm = max(warm, 32)  # the graphs of the usual size buckets are captured (and their GEMMs chosen) in the untimed region
    st = streams[0]
    bufs = batches[:2]

    def enqueue(i):
        step, seeds = seeds_of(i)
        sampler.run_batch(i, seeds, step, bufs[i % 2], table, feat, label, stream=st)

    phases = {"wait_for_batch": 0.0, "enqueue_next_batch": 0.0, "launch_step": 0.0, "wait_for_step": 0.0}

    def region(first, n):
        torch.cuda.synchronize()
        for k in phases:
            phases[k] = 0.0
        t0 = time.perf_counter()
        enqueue(first)
        for j in range(n):
            bt = bufs[(first + j) % 2]
            ta = time.perf_counter()
            m = bt.wait()
            assert not m.overflow
            tb = time.perf_counter()
            # the batch's tensors are read by the step below; the next batch goes to the OTHER buffer
            if j + 1 < n:
                enqueue(first + j + 1)
            tc = time.perf_counter()
            phases["wait_for_batch"] += tb - ta
            phases["enqueue_next_batch"] += tc - tb
            if stepper is not None:
                stepper.step(bt, CooBlock)
            else:
                blocks = []
                for l in range(L):
                    row, col, nsrc, ndst = bt.graph(l)
                    blocks.append(CooBlock(row, col, nsrc, ndst))
                loss = loss_fcn(model(blocks, bt.feat()), bt.label())
                opt.zero_grad()
                loss.backward()
                opt.step()
            td = time.perf_counter()
            torch.cuda.current_stream().synchronize()
            phases["launch_step"] += td - tc
            phases["wait_for_step"] += time.perf_counter() - td
        return time.perf_counter() - t0

    region(next_seq, warm)  # untimed: GEMM kernel selection, optimizer state, lazily loaded code objects
    next_seq += warm
    # A size bucket first met INSIDE the timed region is captured there (~1.5 ms each, seconds with GEMM tuning): such a
    # region is not the steady state the field reports -- it is run again (twice at most), the count is in the line
    attempts = late = 0
    while True:
        attempts += 1
        if stepper and attempts == 3:
            stepper.tune_gemms = False  # the last try: a late bucket is captured with the picks known
        g0 = len(stepper.graphs) if stepper else 0
        with no_gc():
            dt = region(next_seq, timed)
        next_seq += timed
        late = len(stepper.graphs) - g0 if stepper else 0
        if not late or attempts == 3:
            break
    tuned = stepper.tuned_shapes if stepper else 0
    return {"ms_per_step": dt / timed * 1e3, "steps": timed, "timed_regions_run": attempts,
            "graphs_captured_inside_the_reported_region": late,
            "host_ms_per_step": {k: v / timed * 1e3 for k, v in phases.items()},
            "step": ("captured HIP graph per (batch buffer, size bucket): %d graphs, %d replays, %d eager steps"
                     % (len(stepper.graphs), stepper.replays, stepper.eager_steps)) if stepper else "eager (op by op)",
            "gemm_tuning": ("PyTorch TunableOp chose the rocBLAS / hipBLASLt kernel of every GEMM shape before %d size "
                            "buckets were captured (outside the reported region)" % tuned) if tuned else "library defaults",
            "what": "sample + extract of batch k+1 on a side stream under the GraphSAGE step of batch k (examples/models.py: "
                    f"{L} fused SAGEConv layers, hidden 256, fp32, fused Adam; aggregation by fgnn_block_aggregate), one "
                    "GPU"}, next_seq


def default_samplers(n_gpus):
    """1S+1T at 2 GPUs, 2S+6T at 8 (exp/table4/run.py:329-330 for GraphSAGE / papers100M); one sampler below 8"""
    return max(1, n_gpus // 4)


def choose_samplers(world, t_sampler_ms, t_trainer_ms):
    """S of 1 .. world-1 minimising the pipeline's time per batch max(t_s / S, t_t / (world - S)) for the measured
    per-process rates (a sampler process alone, a trainer process alone); ties go to fewer samplers.  The reference
    tunes S per workload by hand (exp/table4/README.md:79-90: 4S / 2S / 2S / 1S).  Returns (S, {S: predicted ms})."""
    pred = {s: max(t_sampler_ms / s, t_trainer_ms / (world - s)) for s in range(1, world)}
    best = min(pred, key=lambda s: (pred[s], s))
    return best, pred


class _FileBarrier:
    """barrier between processes that share nothing but a directory"""

    def __init__(self, d, me, n):
        self.dir, self.me, self.n, self.round = d, me, n, 0

    def wait(self, limit=600.0):
        tag = "cal%d." % self.round
        self.round += 1
        open(os.path.join(self.dir, tag + self.me), "w").close()
        t0 = time.time()
        while len([f for f in os.listdir(self.dir) if f.startswith(tag)]) < self.n:
            if time.time() - t0 > limit:
                raise RuntimeError("calibration barrier %s: only %s arrived" % (tag, sorted(os.listdir(self.dir))))
            time.sleep(0.002)


def run_calibrate_child():
    """--samplers auto: one role of a 1S+1T arch5 job of its own (named regions, the job's dataset) in a child process
    that rank 0 (sampler) / rank 1 (trainer) started before touching the GPU.  The sampler child fills the queue
    ALONE (nobody consumes: warm + K batches, fewer than the queue has slots), then the trainer child drains it ALONE
    -- each stage's own time per batch, the --decoupled measurement in miniature.  Request: one JSON line on stdin
    (an empty line: not needed); answer: one JSON line on stdout."""
    line = sys.stdin.readline()
    if not line.strip():
        return
    req = json.loads(line)
    for k, v in req["env"].items():
        os.environ[k] = v
    import samgraph.torch as sam
    torch.cuda.set_device(req["dev_id"])
    ctx = "cuda:%d" % req["dev_id"]
    warm, K = req["warm"], req["steps"]
    spe = req["steps_per_epoch"]
    cfg = dict(dataset_path=req["dir"], _arch=sam.kArch5, _sample_type=sam.sample_types[req["sample_type"]],
               batch_size=req["batch_size"], num_epoch=(warm + K + spe - 1) // spe + 1,
               _cache_policy=sam.cache_policies["pre_sample"], presample_epoch=req["presample_epochs"],
               cache_percentage=req["cache_ratio"], max_sampling_jobs=10, max_copying_jobs=2, omp_thread_num=8,
               num_sample_worker=1, num_train_worker=1, num_fanout=len(req["fanout"]), fanout=req["fanout"],
               seed=req["seed"])
    sam.config(cfg)
    sam.data_init()
    bar = _FileBarrier(req["sync_dir"], req["role"], 2)
    bar.wait()  # both children have attached to every region
    now = lambda: time.clock_gettime(time.CLOCK_MONOTONIC)  # noqa: E731
    if req["role"] == "s":
        sam.sample_init(0, ctx)  # pre-samples: the ranking the trainer's cache is built from
        bar.wait()
        bar.wait()  # the trainer has built its cache
        for _ in range(warm):
            sam.sample_once()
        sam.get_log_step_value((warm - 1) // spe, (warm - 1) % spe, sam.kLogL1NumSample)  # everything so far published
        with no_gc():
            t0 = now()
            for _ in range(K):
                sam.sample_once()
            last = warm + K - 1
            sam.get_log_step_value(last // spe, last % spe, sam.kLogL1NumSample)  # ... the K timed ones too
            ms = (now() - t0) / K * 1e3
        bar.wait()  # the queue holds warm + K batches
        bar.wait()  # drained
    else:
        bar.wait()
        sam.train_init(0, ctx)
        bar.wait()
        bar.wait()
        sam.extract_start(warm + K)
        for _ in range(warm):
            sam.get_next_batch()
        with no_gc():
            t0 = now()
            for _ in range(K):
                sam.get_next_batch()
            ms = (now() - t0) / K * 1e3
        bar.wait()
    print(json.dumps({"role": req["role"], "ms_per_batch": ms, "steps": K, "warm": warm}), flush=True)
    sam.shutdown()


def calibrate_roles(args, w, dist, rank, world, dev_id, job, steps_per_epoch, cal_child, warm=16, steps=96):
    """collective over the job's ranks: ranks 0 and 1 drive their calibration children (run_calibrate_child), rank 0
    chooses S from the two measured rates and every rank learns it.  A calibration that fails costs the choice, not the
    job: the reference's split is used and the line says why.  --rehearse: no GPU, the rates are --rehearse-rates."""
    mine = None
    if args.rehearse:
        rates = [float(x) for x in args.rehearse_rates.split(",")]
        mine = {"role": "s" if rank == 0 else "t", "ms_per_batch": rates[0] if rank == 0 else rates[1]} if rank < 2 else None
    elif cal_child is not None:
        sync = os.path.join(job["dir"], "calibration_sync")
        os.makedirs(sync, exist_ok=True)
        req = {"env": {"SAMGRAPH_SHM_PREFIX": job["prefix"] + "_cal", "SAMGRAPH_SHM_KEEP": "1",
                       "SAMGRAPH_EMPTY_FEAT": str(args.empty_feat_bits),
                       "SAMGRAPH_LOG_LEVEL": os.environ.get("SAMGRAPH_LOG_LEVEL", "warn")},
               "role": "s" if rank == 0 else "t", "dev_id": dev_id, "dir": job["dir"], "sync_dir": sync, "warm": warm,
               "steps": steps, "steps_per_epoch": steps_per_epoch, "sample_type": args.sample_type,
               "batch_size": w["batch_size"], "cache_ratio": args.cache_ratio,
               "presample_epochs": max(1, args.presample_epochs), "fanout": w["fanout"], "seed": args.seed}
        try:
            o, _ = cal_child.communicate((json.dumps(req) + "\n").encode(),
                                         timeout=float(os.environ.get("FGNN_BENCH_CAL_TIMEOUT", "600")))
            lines = [ln for ln in o.decode(errors="replace").splitlines() if ln.startswith("{")]
            mine = json.loads(lines[-1]) if lines else {"error": "calibration child: rc %s" % cal_child.returncode}
        except Exception as e:
            cal_child.kill()
            mine = {"error": "%s: %s" % (type(e).__name__, e)}
    got = [None] * world
    dist.all_gather_object(got, mine)
    if rank == 0 and not args.rehearse and os.path.isdir("/dev/shm"):
        for f in os.listdir("/dev/shm"):
            if f.startswith(job["prefix"] + "_cal"):
                try:
                    os.unlink(os.path.join("/dev/shm", f))
                except OSError:
                    pass
    ts = next((g["ms_per_batch"] for g in got if g and g.get("role") == "s" and "ms_per_batch" in g), None)
    tt = next((g["ms_per_batch"] for g in got if g and g.get("role") == "t" and "ms_per_batch" in g), None)
    ref = default_samplers(world)
    if ts is None or tt is None:  # (every rank sees the same list: the same decision everywhere)
        return ref, {"mode": "auto: calibration failed, the reference's split is used", "reference_split": ref,
                     "errors": [g.get("error") for g in got if g and "error" in g]}
    S, pred = choose_samplers(world, ts, tt)
    return S, {"mode": "auto", "sampler_ms_per_batch_alone": ts, "trainer_ms_per_batch_alone": tt,
               "predicted_ms_per_batch_by_samplers": {str(k): v for k, v in pred.items()}, "chosen": S,
               "reference_split": ref, "batches_timed": steps,
               "note": "a sampler process alone (queue filling, nobody consuming) and a trainer process alone (draining "
                       "it) as a 1S+1T job of their own before the roles are given out; S = argmin max(t_s / S, "
                       "t_t / (N - S)); --samplers <n> overrides"}


def split_count(total, parts, index):
    """how many of `total` units part `index` of `parts` takes (the trainers' share, multi_gpu/train_graphsage.py:293-298)"""
    return total // parts + (1 if index < total % parts else 0)


def train_region_batches(steps, train_steps, trainers):
    """(warm-up, timed) batches of the region with a training step per batch.  The trainers all-reduce their gradients
    every step, so each of them must take the SAME number of batches -- a remainder would leave the trainers with one
    batch more waiting for the others' all-reduce forever: both counts are multiples of the trainer count (the
    reference pads the train set to equal shares for the same reason, dist_shuffler_aligned.cc:50-59)."""
    t = max(trainers, 1)
    timed = max(min(steps, train_steps) // t, 1) * t
    warm = max(min(8, timed) // t, 1) * t
    return warm, timed


def span_margins(warmup, trainers, decoupled=False):
    """(lead, tail) batches around the timed windows of a pipeline span: consumed, stamped, not counted.  The lead is the
    warm-up AND the queue's transient: the span starts on an empty queue, and where the trainers are the slower side
    (every default role split at cache 0.2) the steady state is a FULL queue -- reached after about
    slots x t_sample / (t_extract - t_sample) consumed batches, ~70 at 1S+1T and ~320 at 2S+6T; twice the queue's slots
    covers both (340 batches of 0.05-0.45 ms each).  On separate GPUs the fill level does not change the trainers' rate;
    where sampler and trainer SHARE one GPU (the development box) it does: 0.34 ms per batch while the queue fills,
    0.44 once it is full (profiles/r05_b_windows_transient.txt).  tail: two batches per trainer cover the spread of the
    trainers' finishing times."""
    if decoupled:
        return 1, 0
    return max(warmup, 2 * trainers, 2 * QUEUE_SLOTS), max(warmup, 2 * trainers)


def span_total(lead, windows, steps, tail, trainers, train):
    """batches of one span; with a training step per batch every trainer must take the same number (all-reduce)"""
    total = lead + windows * steps + tail
    return (total + trainers - 1) // trainers * trainers if train else total


def read_windows(stamps, lead, windows, steps):
    """stamps: [(t, key)] of every consumed batch of a span, any order.  Returns (merged, [(t_begin, t_end, keys)] per
    window): window j = the batches lead + j*steps .. lead + (j+1)*steps - 1 in consumption order, its clock runs from
    the stamp of the batch consumed just before it to the stamp of its last batch."""
    merged = sorted(stamps)
    out = []
    for j in range(windows):
        a = lead + j * steps
        out.append((merged[a - 1][0], merged[a + steps - 1][0], [k for _, k in merged[a:a + steps]]))
    return merged, out


def pipeline_roles(world, samplers=None):
    """(samplers, trainers); samplers: a count, or None / 0 / "auto" before the choice is made = the reference's split"""
    s = int(samplers) if samplers and samplers != "auto" else default_samplers(world)
    if not (0 < s < world):
        raise ValueError("need at least one sampler and one trainer: %d samplers of %d ranks" % (s, world))
    return s, world - s


def write_dataset(args, w, dev, out_dir):
    """the engine's on-disk layout without feat.bin (SAMGRAPH_EMPTY_FEAT, like papers100M_empty)"""
    os.makedirs(out_dir, exist_ok=True)
    indptr, indices, ne, desc = gen_graph(args, w, dev)
    indptr.cpu().numpy().view(np.uint32).tofile(os.path.join(out_dir, "indptr.bin"))
    chunk = 1 << 28
    with open(os.path.join(out_dir, "indices.bin"), "wb") as f:
        for a in range(0, ne, chunk):
            f.write(indices[a:a + chunk].cpu().numpy().view(np.uint32).tobytes())
    del indptr, indices
    from fgnn_hip import rmat
    train = rmat.train_set(w["num_node"], w["num_train"], 1, dev)
    train.cpu().numpy().view(np.uint32).tofile(os.path.join(out_dir, "train_set.bin"))
    np.zeros(0, dtype=np.uint32).tofile(os.path.join(out_dir, "valid_set.bin"))
    np.zeros(0, dtype=np.uint32).tofile(os.path.join(out_dir, "test_set.bin"))
    with open(os.path.join(out_dir, "meta.txt"), "w") as f:
        f.write(f"NUM_NODE {w['num_node']}\nNUM_EDGE {ne}\nFEAT_DIM {w['feat_dim']}\nNUM_CLASS {w['num_class']}\n"
                f"NUM_TRAIN_SET {w['num_train']}\nNUM_VALID_SET 0\nNUM_TEST_SET 0\n")
    torch.cuda.empty_cache()
    return ne, desc


class EngineBackend:
    """the product: arch5 through samgraph.torch / c_lib.so on this rank's GPU"""

    def __init__(self, args, w, job, S, T, is_sampler, idx, dev_id, num_epoch):
        import samgraph.torch as sam
        self.sam, self.w, self.is_sampler, self.idx, self.ctx = sam, w, is_sampler, idx, "cuda:%d" % dev_id
        self.dev_id = dev_id
        cfg = dict(dataset_path=job["dir"], _arch=sam.kArch5, _sample_type=sam.sample_types[args.sample_type],
                   batch_size=w["batch_size"], num_epoch=num_epoch, _cache_policy=sam.cache_policies["pre_sample"],
                   presample_epoch=max(1, args.presample_epochs), cache_percentage=args.cache_ratio,
                   max_sampling_jobs=10, max_copying_jobs=2, omp_thread_num=8, num_sample_worker=S, num_train_worker=T, num_fanout=len(w["fanout"]),
                   fanout=w["fanout"], seed=args.seed)
        sam.config(cfg)
        sam.data_init()  # attaches to / creates the job's shared regions; no GPU touched

    def role_init(self):
        torch.cuda.set_device(self.dev_id)
        if self.is_sampler:
            self.sam.sample_init(self.idx, self.ctx)  # sampler 0 pre-samples; the others wait for it inside
        else:
            self.sam.train_init(self.idx, self.ctx)

    def num_local_step(self):
        return self.sam.num_local_step()

    def sample_once(self):
        self.sam.sample_once()

    def extract_start(self, n):
        self.sam.extract_start(n)

    def next_batch(self):
        return self.sam.get_next_batch()

    def blocks(self, key):
        return self.sam.get_dgl_blocks(key, len(self.w["fanout"]))

    def sampler_stats(self, keys):
        sam = self.sam
        return {"edges": sum(sam.get_log_step_value(e, s, sam.kLogL1NumSample) for e, s in keys)}

    def trainer_stats(self, keys):
        sam, row_b = self.sam, self.w["feat_dim"] * 4

        def tot(item):
            return sum(sam.get_log_step_value(e, s, item) for e, s in keys)
        return {"rows": tot(sam.kLogL1FeatureBytes) / row_b, "miss_rows": tot(sam.kLogL1MissBytes) / row_b,
                "graph_bytes": tot(sam.kLogL1GraphBytes), "ms_miss": tot(sam.kLogL3CacheCombineMissTime) * 1e3,
                "ms_cache": tot(sam.kLogL3CacheCombineCacheTime) * 1e3}

    def queue_stats(self, rings):
        return [self.sam.ext_queue_stats(r) for r in range(rings)]

    def ring_mappings(self, rings):
        """per sampler ring: how THIS process read its payloads (samgraph_ext_ring_mapping)"""
        return [self.sam.ext_ring_mapping(r) for r in range(rings)]

    def shutdown(self):
        self.sam.shutdown()


class RehearsalBackend:
    """--rehearse: the job's control plane without a GPU -- launcher, rendezvous, roles, step ranges, the REAL shared
    ring of the engine (its host-only hooks library) between the rank processes (named regions), reductions and the JSON line; a batch is an empty
    message {key, a number of edges derived from the key}.  Numbers printed in this mode measure nothing."""
    SLOTS, SLOT_BYTES = 8, 4096

    def __init__(self, args, w, job, S, T, is_sampler, idx, dev_id, num_epoch):
        import ctypes as C
        self.C = C
        self.eng = C.CDLL(os.path.join(ROOT, "fgnn-artifacts_amd", "samgraph", "torch", "fgnn_engine_hooks.so"))
        self.eng.fgnn_host_queue_open.restype = C.c_void_p
        self.q = C.c_void_p(self.eng.fgnn_host_queue_open(C.c_size_t(self.SLOTS), C.c_size_t(self.SLOT_BYTES)))
        bs = w["batch_size"]
        self.steps_per_epoch = (w["num_train"] + bs - 1) // bs
        self.first, self.local = local_step_range(self.steps_per_epoch, idx, S) if is_sampler else (0, 0)
        self.j = 0
        self.got = {}

    @staticmethod
    def edges_of(key):
        return 1000 + key % 97

    def role_init(self):
        pass

    def num_local_step(self):
        return self.local

    def sample_once(self):
        key = (self.j // self.local) * self.steps_per_epoch + self.first + self.j % self.local
        self.j += 1
        self.eng.fgnn_host_queue_send(self.q, self.C.c_uint64(key), self.C.c_uint64(self.edges_of(key)))

    def extract_start(self, n):
        pass

    def next_batch(self):
        k, v = self.C.c_uint64(), self.C.c_uint64()
        self.eng.fgnn_host_queue_recv(self.q, self.C.byref(k), self.C.byref(v))
        self.got[k.value] = v.value
        return k.value

    def sampler_stats(self, keys):
        return {"edges": float(sum(self.edges_of(e * self.steps_per_epoch + s) for e, s in keys))}

    def trainer_stats(self, keys):
        assert all(self.got[e * self.steps_per_epoch + s] == self.edges_of(e * self.steps_per_epoch + s) for e, s in keys)
        return {"rows": float(len(keys)), "miss_rows": 0.0, "graph_bytes": 0.0, "ms_miss": 0.0, "ms_cache": 0.0}

    def queue_stats(self, rings):
        return [None] * rings

    def ring_mappings(self, rings):
        return [None] * rings

    def shutdown(self):
        self.eng.fgnn_host_queue_close(self.q)


def run_pipeline_rank(args, rank, world):
    import datetime
    import shutil
    if os.environ.get("FGNN_BENCH_WATCHDOG"):  # a stuck rank shows where it is stuck, then exits
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["FGNN_BENCH_WATCHDOG"]), exit=True)
    import torch.distributed as dist
    dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=900))
    n_dev = torch.cuda.device_count()
    if n_dev == 0 and not args.rehearse:
        sys.exit("bench.py needs a GPU: the HIP path has no CPU fallback")
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    dev_id = local_rank % max(n_dev, 1)  # more ranks than GPUs (a functional check on one GPU): they share
    n1_child = None
    exit_msg = None
    child_env = {k: v for k, v in os.environ.items()
                 if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK",
                              "TORCHELASTIC_RUN_ID", "FGNN_BENCH_CHILD")}
    if rank == 0 and not (args.rehearse or args.no_n1_point or args.decoupled):
        # started now, before this process touches the GPU; it sleeps on its stdin until the spans are done
        # FGNN_BENCH_N1_WRAP (tools): a profiler in front of the child, e.g. "rocprofv3 --kernel-trace --stats -d DIR --"
        wrap = os.environ.get("FGNN_BENCH_N1_WRAP", "").split()
        n1_child = subprocess.Popen(wrap + [sys.executable, os.path.abspath(__file__), "--n1-point-child"], env=child_env,
                                    stdin=subprocess.PIPE, stdout=subprocess.PIPE)
    # --samplers auto (the default): how many of the ranks sample is chosen from MEASURED rates -- a sampler process
    # alone and a trainer process alone, a few dozen batches each, in two child processes of ranks 0 and 1 (a 1S+1T job
    # of their own over the job's dataset; started now, before anything here touches the GPU).  Two ranks leave no choice
    auto = str(args.samplers).lower() in ("auto", "0", "none")
    calibrate = auto and world >= 3 and not args.decoupled
    cal_child = None
    if calibrate and not args.rehearse and rank in (0, 1):
        cal_child = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--calibrate-child"], env=child_env,
                                     stdin=subprocess.PIPE, stdout=subprocess.PIPE)
    w = WORKLOADS[args.workload]
    if args.num_walks and "num_walks" in w:
        w = dict(w, num_walks=args.num_walks)
    if args.sample_type is None:
        args.sample_type = w["sample_type"]
    bs = w["batch_size"]
    W, K = args.warmup, args.steps
    steps_per_epoch = (w["num_train"] + bs - 1) // bs
    # ---- job-wide names from rank 0: shared-memory prefix (the processes have no common forking parent) and the
    # dataset directory
    obj = [None]
    if rank == 0:
        tag = "fgnn_bench_%d_%x" % (os.getpid(), int(time.time() * 1e3) & 0xFFFFFF)
        base = "/dev/shm" if os.path.isdir("/dev/shm") and \
            os.statvfs("/dev/shm").f_bavail * os.statvfs("/dev/shm").f_frsize > (48 << 30) else "/tmp"
        obj[0] = {"prefix": tag, "dir": os.path.join(base, tag + "_ds")}
    dist.broadcast_object_list(obj, 0)
    job = obj[0]
    os.environ["SAMGRAPH_SHM_PREFIX"] = job["prefix"]
    os.environ["SAMGRAPH_SHM_KEEP"] = "1"  # rank 0 removes the names after the last barrier
    os.environ["SAMGRAPH_EMPTY_FEAT"] = str(args.empty_feat_bits)
    os.environ.setdefault("SAMGRAPH_LOG_LEVEL", "warn")
    t_setup = time.time()
    info = [None]
    if rank == 0:
        if args.rehearse:
            info[0] = {"num_edge": w["num_edge"], "graph": "none (rehearsal)"}
        else:
            torch.cuda.set_device(dev_id)
            ne, desc = write_dataset(args, w, torch.device("cuda", dev_id), job["dir"])
            info[0] = {"num_edge": ne, "graph": desc}
    dist.broadcast_object_list(info, 0)
    # ---- roles
    sampler_choice = {"mode": "fixed (--samplers)" if not auto else "auto: two ranks leave no choice" if world < 3
                      else "auto switched off by --decoupled (a per-stage diagnostic run)",
                      "reference_split": default_samplers(world)}
    S = pipeline_roles(world, None if auto else args.samplers)[0]
    if calibrate:
        S, sampler_choice = calibrate_roles(args, w, dist, rank, world, dev_id, job, steps_per_epoch, cal_child)
        cal_child = None
    T = world - S
    is_sampler = rank < S
    idx = rank if is_sampler else rank - S
    # Steady-state timing (no barrier inside the measured span): one SPAN of lead + R x K + tail batches goes through the
    # pipeline with the samplers free-running (bounded by the ring) and every trainer stamping CLOCK_MONOTONIC -- one
    # node, one clock for all ranks -- when a batch has been consumed; rank 0 merges the stamps and reads R back-to-back
    # windows of K consecutively consumed batches out of the middle.  `lead` covers the pipeline's fill (first message =
    # one sample chain + one extract) and the warm-up, `tail` the drain (trainers finishing their shares at slightly
    # different times).  The reference times the same loop per epoch (multi_gpu/train_graphsage.py:286-330).
    R = 1 if args.decoupled else max(1, args.windows)
    T_ = T
    lead, tail = span_margins(W, T_, args.decoupled)
    K2 = 0 if (args.no_train_leg or args.rehearse) else train_region_batches(K, args.train_steps, T_)[1]
    total1 = span_total(lead, R, K, tail, T_, False)
    total2 = span_total(lead, R, K2, tail, T_, True) if K2 else 0
    try:
        min_local = steps_per_epoch // S
        per_sampler = max(sum(split_count(n, S, 0) for n in (total1, total2)), 1)
        num_epoch = (per_sampler + min_local - 1) // min_local + 1
        # hand-off self-check at the start of the span's lead: every sampler checksums its first W // S messages, the
        # trainer that receives one recomputes the sum through the address it reads the payload from (the sampler's
        # HBM slot mapped over xGMI, or the pinned host slot) and the job dies on a mismatch (eng_engine.cc)
        check_n = max(W, 1) // S if W // S else 1  # a few: the receiver verifies synchronously (a long checked lead
        # would hold the trainers back and delay the steady state the windows are read from)
        os.environ["SAMGRAPH_HANDOFF_CHECK"] = str(check_n)
        be = (RehearsalBackend if args.rehearse else EngineBackend)(args, w, job, S, T, is_sampler, idx, dev_id, num_epoch)
        dist.barrier()  # every process has attached to every shared region
        if is_sampler:
            be.role_init()
            dist.barrier()  # the rank list is in shared memory: trainers may build their caches
        else:
            dist.barrier()
            be.role_init()
        dist.barrier()
        t_setup = time.time() - t_setup

        first_step, local_steps = local_step_range(steps_per_epoch, idx, S) if is_sampler else (0, 0)
        if is_sampler:
            assert be.num_local_step() == local_steps, (be.num_local_step(), local_steps)
        sampled = [0]  # batches this sampler has produced
        keys = []

        links, rccl_ok = link_selftest(dist, rank, world, dev_id, n_dev, args.rehearse)
        if not rccl_ok:  # every rank has the same verdict: no region that needs RCCL
            K2, total2 = 0, 0
        model = opt = loss_fcn = None
        if K2 and T > 1:
            # gradient all-reduce between the trainers: RCCL ("nccl") when each has its own GPU, gloo when ranks
            # share one (functional check on a single-GPU box: RCCL refuses two ranks on one device).  new_group is
            # a collective call: every rank takes part
            tgroup = dist.new_group(ranks=list(range(S, world)), backend="nccl" if n_dev >= world else "gloo",
                                    timeout=datetime.timedelta(seconds=600))
        if not is_sampler and K2:
            sys.path.insert(0, os.path.join(ROOT, "examples"))
            from models import MODELS
            model = MODELS["graphsage"](w["feat_dim"], 256, w["num_class"], len(w["fanout"]), 0.5).to("cuda:%d" % dev_id)
            if T > 1:
                model = torch.nn.parallel.DistributedDataParallel(
                    model, device_ids=[dev_id] if n_dev >= world else None, process_group=tgroup)
            loss_fcn = torch.nn.CrossEntropyLoss()
            from fgnn_hip.nn import softmax_xent
            from fgnn_hip.nn import Adam as FusedAdam
            opt = FusedAdam(model.parameters(), lr=0.003)
            (model.module if T > 1 else model).dropout_step = opt.step_count
            model.train()

        now = lambda: time.clock_gettime(time.CLOCK_MONOTONIC)  # noqa: E731  (one node: every rank reads the same clock)

        def span(total, train):
            """`total` batches through the pipeline with NO barrier between the first and the last: this rank's share as
            a sampler (sample_once, free-running against the ring) or as a trainer (get_next_batch [+ training step],
            one CLOCK_MONOTONIC stamp per consumed batch).  Returns (stamps [(t, key)], seconds in this rank's loop)."""
            mine = split_count(total, S, idx) if is_sampler else split_count(total, T, idx)
            stamps = []
            import gc
            gc.collect()
            gc.disable()  # (no_gc: a collector pass inside a rank's loop is a multi-millisecond hole in the stamps)
            dist.barrier()
            t0 = now()
            if is_sampler:
                for _ in range(mine):
                    be.sample_once()
                    j = sampled[0]
                    sampled[0] += 1
                    keys.append((j // local_steps, first_step + j % local_steps))
                loop_s = now() - t0
                if args.decoupled:
                    time.sleep(0.02)  # the publisher thread publishes the last batches as their GPU work completes
                    dist.barrier()
            else:
                if args.decoupled:
                    dist.barrier()  # diagnostic: the samplers have filled the queue, the trainers run alone
                    t0 = now()
                if mine:
                    be.extract_start(mine)
                for _ in range(mine):
                    key = be.next_batch()
                    if train:
                        blocks, feat, label = be.blocks(key)
                        out = model(blocks, feat)
                        loss, g = softmax_xent(out, label)  # CrossEntropyLoss + its gradient, one launch
                        opt.zero_grad()
                        out.backward(g)
                        opt.step()
                        torch.cuda.current_stream().synchronize()
                    stamps.append((now(), key))
                    keys.append((key // steps_per_epoch, key % steps_per_epoch))
                loop_s = now() - t0
            gc.enable()
            dist.barrier()  # every batch of the span has been consumed
            return stamps, loop_s

        def collect(stamps, per_key):
            """every rank's stamps and per-batch figures on rank 0 (after the span: nothing of this is timed)"""
            got = [None] * world
            dist.all_gather_object(got, (stamps, per_key))
            all_stamps = [x for st, _ in got for x in st]
            merged_pk = {}
            for _, pk in got:
                for k, v in pk.items():
                    merged_pk.setdefault(k, {}).update(v)
            return all_stamps, merged_pk

        # ---- span 1: sample -> hand-off -> cached extraction
        stamps, loop_s = span(total1, False)
        # a sampler's publisher thread logs a batch when it publishes it: all published by the barrier above
        per_key = {}
        for e, st in keys:
            g = e * steps_per_epoch + st
            per_key[g] = ({"edges": be.sampler_stats([(e, st)])["edges"]} if is_sampler
                          else be.trainer_stats([(e, st)]))
        all_stamps, pk = collect(stamps, per_key)
        n_produced = len(keys) if is_sampler else 0
        del keys[:]
        # ---- span 2: the same with a training step per consumed batch
        train_stamps = []
        if K2:
            stamps2, _ = span(total2, True)
            train_stamps, _ = collect(stamps2, {})
            del keys[:]

        def red(vals, op):
            t = torch.tensor(vals, dtype=torch.float64)
            dist.all_reduce(t, op=op)
            return [float(x) for x in t]
        s_loop, t_loop, setup_max = red([loop_s if is_sampler else 0.0, loop_s if not is_sampler else 0.0, t_setup],
                                        dist.ReduceOp.MAX)
        nb_s, = red([n_produced], dist.ReduceOp.SUM)
        dist.barrier()  # every trainer has verified what it was going to verify
        rings = be.queue_stats(S) if rank == 0 else None  # shared counters: any process of the job can read them
        # where every rank's GPU hangs (NUMA node of its PCIe root) next to where the shared host feature table lives
        gnodes = [None] * world
        dist.all_gather_object(gnodes, None if args.rehearse else gpu_numa_node(dev_id))
        # every trainer: how it read each sampler's ring (mapped device to device, or copied back through the host slot)
        maps = [None] * world
        dist.all_gather_object(maps, None if is_sampler else {"rank": rank, "device": dev_id,
                                                             "rings": be.ring_mappings(S)})
        be.shutdown()
        dist.barrier()
        n1_point = None
        if rank == 0:
            n1_point = {"value": None, "why": "control-plane rehearsal" if args.rehearse else "not requested"}
            if n1_child is not None:
                req = {"env": {"SAMGRAPH_SHM_PREFIX": job["prefix"] + "_n1", "SAMGRAPH_EMPTY_FEAT": str(args.empty_feat_bits),
                               "SAMGRAPH_LOG_LEVEL": os.environ.get("SAMGRAPH_LOG_LEVEL", "warn")},
                       "dev_id": dev_id, "dir": job["dir"], "lead": lead, "windows": R, "steps": K, "tail": tail,
                       "steps_per_epoch": steps_per_epoch, "sample_type": args.sample_type, "batch_size": bs,
                       "cache_ratio": args.cache_ratio, "presample_epochs": max(1, args.presample_epochs),
                       "fanout": w["fanout"], "seed": args.seed,
                       "row_bytes": w["feat_dim"] * 4}
                try:
                    o, _ = n1_child.communicate((json.dumps(req) + "\n").encode(), timeout=float(
                        os.environ.get("FGNN_BENCH_N1_TIMEOUT", "300")))
                    lines = [ln for ln in o.decode(errors="replace").splitlines() if ln.startswith("{")]
                    n1_point = json.loads(lines[-1]) if lines else {"value": None, "error": "rc %s" % n1_child.returncode}
                except Exception as e:  # a secondary measurement must not cost the line
                    n1_child.kill()
                    n1_point = {"value": None, "error": "%s: %s" % (type(e).__name__, e)}
                n1_child = None
        if rank == 0:
            assert int(nb_s) == total1 and len(all_stamps) == total1, (nb_s, len(all_stamps), total1)
            assert len({k for _, k in all_stamps}) == total1  # every batch reached exactly one trainer
            merged, wins = read_windows(all_stamps, lead, R, K)
            if os.environ.get("FGNN_BENCH_DUMP_STAMPS"):  # tools: every consumed batch's stamp (seconds from the first)
                with open(os.environ["FGNN_BENCH_DUMP_STAMPS"], "w") as f:
                    f.write("# first stamp at CLOCK_MONOTONIC %.6f\n" % merged[0][0])
                    for t, k in merged:
                        f.write("%.6f %d\n" % (t - merged[0][0], k))
            win_ms = [(t1 - t0_) / K * 1e3 for t0_, t1, _ in wins]
            order = sorted(range(R), key=lambda r: win_ms[r])
            med = order[(R - 1) // 2]  # the median window (the slower of the middle two for an even R)
            t_max = wins[med][1] - wins[med][0]
            mkeys = wins[med][2]
            edges = sum(pk[k]["edges"] for k in mkeys)
            rows, miss_rows, graph_bytes = (sum(pk[k][n] for k in mkeys) for n in ("rows", "miss_rows", "graph_bytes"))
            # launch averages (roofline): over the launches of ALL R windows (R x K of each kind)
            wkeys = [k for _, _, ks in wins for k in ks]
            rows_a, miss_rows_a, ms_miss, ms_cache = (sum(pk[k][n] for k in wkeys)
                                                      for n in ("rows", "miss_rows", "ms_miss", "ms_cache"))
            n_launch = len(wkeys)
            t_train = train_win_ms = None
            if K2:
                _, twins = read_windows(train_stamps, lead, R, K2)
                train_win_ms = [(t1 - t0_) / K2 * 1e3 for t0_, t1, _ in twins]
                t_train = sorted(train_win_ms)[(R - 1) // 2] * K2 * 1e-3
            live = [r for r in rings if r]
            handoff = {"rings": rings, "check_messages_per_sampler": check_n,
                       "verified": sum(r["verified"] for r in live), "check_failed": sum(r["check_failed"] for r in live),
                       "sent_device": sum(r["sent_device"] for r in live), "sent_host": sum(r["sent_host"] for r in live),
                       "spilled": sum(r["spilled"] for r in live),
                       "transport": ("none (rehearsal)" if not live else
                                     "sampler HBM ring, peer-read by the trainers" if all(
                                         r["sent_host"] == 0 and r["spilled"] == 0 and r["sent_device"] > 0 for r in live)
                                     else "pinned host ring" if all(r["sent_device"] == 0 for r in live)
                                     else "MIXED: part of the messages fell back to the pinned host ring"),
                       "note": "per sampler ring: slots, messages by payload location, copies back on request, and the "
                               "warm-up messages the receiving trainers verified end to end (a mismatch aborts the job)"}
            handoff["trainers"] = [m for m in maps if m]
            if live and handoff["check_failed"]:
                sys.exit("bench.py: hand-off check failed: %s" % handoff)
            # A run that was meant to read the samplers' HBM rings peer to peer but moved payloads through pinned host
            # memory is a different (slower) system: it must not pass for the real thing.  Asked-for host transport
            # (SAMGRAPH_DEVICE_RING_SLOTS=0, the forced-spill test switch) is fine
            asked_host = os.environ.get("SAMGRAPH_DEVICE_RING_SLOTS") == "0" or \
                os.environ.get("SAMGRAPH_DEVICE_RING_FORCE_SPILL") not in (None, "", "0")
            refused = [(m["rank"], i) for m in handoff["trainers"] for i, g in enumerate(m["rings"]) if g and g["state"] == 3]
            degraded = live and (refused or any(r["sent_host"] or r["spilled"] for r in live))
            handoff["degraded"] = bool(degraded) and not asked_host
            row_b = w["feat_dim"] * 4
            hit_rows = rows - miss_rows
            handoff_bytes = graph_bytes + 8 * rows + 4 * bs * K  # COO arrays + (miss|cache) index pairs + output ids
            cache_launch_bytes = (rows_a - miss_rows_a) * (2 * row_b + 8)  # of all R windows' launches, like ms_cache
            out = {
                "metric": f"sampled-edges/sec ({args.sample_type} fanout {'/'.join(map(str, w['fanout']))}, batch {bs}, "
                          "factored pipeline: sampler GPUs (sample + dedup + remap + cache-index split) -> HBM message "
                          "ring -> trainer GPUs (cached feature extraction)); edges of K consecutively consumed batches "
                          f"/ the time the trainers took to consume them, median of {R} back-to-back windows of a "
                          "continuously full pipeline",
                "value": edges / t_max, "unit": "edges/s", "n_gpus": world, "steps": K, "warmup": W,
                "ms_per_step": t_max / K * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
                "dtype": "u32", "data": "synthetic" if not args.rehearse else "none (control-plane rehearsal)",
                "windows": {"count": R, "ms_per_step": win_ms, "min": min(win_ms), "max": max(win_ms), "median_index": med,
                            "lead_batches": lead, "tail_batches": total1 - lead - R * K, "span_batches": total1,
                            "span_ms_per_step": (wins[-1][1] - wins[0][0]) / (R * K) * 1e3,
                            "median_window_keys": mkeys,
                            "clock": "CLOCK_MONOTONIC stamped by the consuming trainer after every batch; window j = "
                                     "batches lead + jK .. lead + (j+1)K - 1 in consumption order over all trainers, from "
                                     "the stamp of the batch before it to the stamp of its last; no barrier inside the "
                                     "span; value / ms_per_step come from the median window"},
                "config": {"workload": f"{args.workload}-shaped synthetic graph: {info[0]['graph']}; N={w['num_node']}, "
                                       f"E={info[0]['num_edge']}, train set {w['num_train']} uniform random ids (seed 1), "
                                       f"{args.sample_type} fanout {w['fanout']}, batch {bs}; features in host memory "
                                       f"(2^{args.empty_feat_bits} rows, ids masked = SAMGRAPH_EMPTY_FEAT), pre-sample "
                                       f"cache ratio {args.cache_ratio} (presample_epoch {max(1, args.presample_epochs)}) "
                                       "in every trainer's HBM; arch5 through "
                                       "samgraph.torch / c_lib.so, one process per GPU",
                           "global_batch": bs, "parallelism": f"{S}S+{T}T (samplers -> device ring -> trainers)"},
                "roofline": {"bound": "hbm", "kernel": "extract_fused_kernel, HBM band (CombineCacheData on the trainer GPUs; the "
                                                        "same launch's link band pulls the miss rows over the host link)",
                             "achieved": cache_launch_bytes / (ms_cache * 1e-3) / 1e9 if ms_cache else None,
                             "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": cache_launch_bytes / (ms_cache * 1e-3) / 1e9 / HBM_PEAK_GBS if ms_cache else None,
                             "traffic": None, "avg_launch_ms": ms_cache / n_launch, "timed_launches": n_launch,
                             "algorithmic_bytes_per_launch": cache_launch_bytes / n_launch,
                             "note": "hit rows x (row read + row write + 2 index words) / the HBM band's first start .. "
                                     "last end inside the one-launch extraction (device clock stamps of its workgroups), "
                                     "summed over the trainers' batches"},
                "pipeline": {
                    "samplers": S, "trainers": T, "devices": min(n_dev, world), "sampler_choice": sampler_choice,
                    "sampler_loop_ms_per_batch": s_loop / max(split_count(total1, S, 0), 1) * 1e3,
                    "trainer_loop_ms_per_batch": t_loop / max(split_count(total1, T, 0), 1) * 1e3,
                    "loop_note": "wall time of a rank's whole span loop / its batches (MAX within the role): a sampler's "
                                 "includes its waits on a full ring, a trainer's its waits on an empty one -- each "
                                 "stage ALONE only with --decoupled",
                    "sampler_busy_s": s_loop, "trainer_busy_s": t_loop,
                    # steady state of the consuming side: the second half of the span's stamps (the first batches of a
                    # process pay its pool's first allocations; with --decoupled this is the trainers ALONE)
                    "consumed_second_half_ms_per_batch":
                        (merged[-1][0] - merged[len(merged) // 2][0]) / max(len(merged) - 1 - len(merged) // 2, 1) * 1e3,
                    "trainer_rows_per_s": rows / t_max, "hit_rate": hit_rows / max(rows, 1.0),
                    "handoff_bytes_per_step": handoff_bytes / K, "handoff_GBps": handoff_bytes / t_max / 1e9,
                    "handoff_peak_GBps": XGMI_LINK_GBS, "handoff": handoff, "links": links,
                    "numa": {"gpu_node_of_rank": gnodes, "nodes_with_memory": numa_nodes_with_memory(),
                             "host_feat_policy": os.environ.get("SAMGRAPH_HOST_FEAT_NUMA",
                                                                "interleave over the nodes with memory (default)"),
                             "note": "every trainer pulls its miss rows out of ONE shared host table (DESIGN 6)"},
                    "n1_point_of_this_curve": n1_point,
                    "miss": {"bound": "host link", "bytes_per_step": miss_rows * row_b / K,
                             "achieved": miss_rows * row_b / t_max / 1e9 / T, "peak": HOST_LINK_GBS,
                             "unit": "GB/s per trainer GPU", "frac": miss_rows * row_b / t_max / 1e9 / T / HOST_LINK_GBS,
                             "avg_band_ms": ms_miss / n_launch,
                             "band_GBps": miss_rows_a * row_b / (ms_miss * 1e-3) / 1e9 if ms_miss else None,
                             "note": "achieved = miss bytes / wall time per trainer; band = the link band of one launch "
                                     "(up to four batches' bands share a trainer's link)"},
                },
                "epoch_time_s": {"sample_plus_extract": steps_per_epoch * t_max / K,
                                 "with_training": steps_per_epoch * t_train / K2 if K2 else None,
                                 "training_steps_timed": K2, "training_windows_ms_per_step": train_win_ms,
                                 "note": f"{steps_per_epoch} steps/epoch x seconds per step of the median window; with_training "
                                         "= the same pipeline with a GraphSAGE step (examples/models.py, hidden 256, "
                                         "Adam) on every batch, gradients all-reduced over RCCL between trainers"},
                "edges_per_step": edges / K, "input_nodes_per_step": rows / K, "setup_s": setup_max,
            }
            print(json.dumps(out), flush=True)
            if handoff.get("degraded") and n_dev >= world:
                exit_msg = ("bench.py: every rank has its own GPU but payloads went through the pinned host ring "
                            "(trainer, ring) refused: %s; rings: %s -- the line above is NOT the peer-read pipeline"
                            % (refused, rings))
    finally:
        if n1_child is not None:  # never asked (an error above): an empty line ends it
            try:
                n1_child.communicate(b"\n", timeout=30)
            except Exception:
                n1_child.kill()
        try:
            dist.barrier()
        except Exception:
            pass
        if rank == 0:
            shutil.rmtree(job["dir"], ignore_errors=True)
            if os.path.isdir("/dev/shm"):
                for f in os.listdir("/dev/shm"):
                    if f.startswith(job["prefix"]):
                        try:
                            os.unlink(os.path.join("/dev/shm", f))
                        except OSError:
                            pass
    dist.destroy_process_group()
    if exit_msg:
        sys.exit(exit_msg)


def limited_collective(dist, world, body, limit_s):
    """`body` (collective calls that may never return when a peer has failed) on a helper thread; this rank waits for
    ITS OWN thread for at most limit_s seconds, then every rank learns over the job's gloo group who finished.
    Returns (body's result | None, {rank: reason} of the ranks that did not finish -- the same dict on every rank)."""
    import threading
    res = {}

    def run():
        try:
            res["rec"] = body()
        except Exception as e:
            res["err"] = "%s: %s" % (type(e).__name__, e)
    th = threading.Thread(target=run, daemon=True)
    th.start()
    th.join(limit_s)
    mine = None if "rec" in res else res.get("err", "no answer within %.0f s" % limit_s)
    status = [None] * world
    dist.all_gather_object(status, mine)  # gloo: works whatever the helper thread is stuck in
    return res.get("rec"), {r: e for r, e in enumerate(status) if e is not None}


def link_selftest(dist, rank, world, dev_id, n_dev, rehearse, limit_s=None):
    """First-contact proof for N >= 2, run once before the timed span: ONE RCCL all-reduce over ALL ranks (samplers
    included: `rccl_world` == world says RCCL saw every rank; the data path itself has no collective), its bus
    bandwidth on a 64 MiB payload, and the peer-access matrix between the ranks' GPUs (what the trainers' peer reads of
    the samplers' HBM rings rest on).  With fewer GPUs than ranks RCCL refuses (two ranks on one device): recorded as
    such, nothing is faked.  Every rank takes part; rank 0 gets the record.

    A rank whose RCCL initialisation fails ALONE must not leave the others inside a collective: the RCCL calls run on a
    helper thread, every rank waits for ITS OWN thread for at most `limit_s` seconds, then all ranks agree over gloo on
    who finished; on any failure every rank aborts its communicator (ncclCommAbort ends a kernel that waits for a peer)
    and the record says which ranks failed and why.  Returns (record on rank 0 | None, rccl_ok on every rank)."""
    import datetime
    if limit_s is None:
        limit_s = float(os.environ.get("FGNN_BENCH_LINK_TIMEOUT", "150"))
    if rehearse:
        return {"rccl_world": None, "why": "control-plane rehearsal: no GPU work"}, True
    ok = True
    if n_dev < world:
        rec = {"rccl_world": None, "why": "%d ranks share %d GPU(s): RCCL needs a device per rank" % (world, n_dev)}
    else:
        grp = dist.new_group(ranks=list(range(world)), backend="nccl", timeout=datetime.timedelta(seconds=limit_s + 30))
        dev = torch.device("cuda", dev_id)

        def body():
            torch.cuda.set_device(dev)
            if os.environ.get("FGNN_BENCH_LINK_FAIL_RANK") == str(rank):  # tests: this rank fails alone
                raise RuntimeError("injected failure (FGNN_BENCH_LINK_FAIL_RANK)")
            one = torch.ones(1, device=dev)
            dist.all_reduce(one, group=grp)
            torch.cuda.synchronize(dev)
            buf = torch.ones(16 << 20, dtype=torch.float32, device=dev)  # 64 MiB
            dist.all_reduce(buf, group=grp)  # first use of the size
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(5):
                dist.all_reduce(buf, group=grp)
            torch.cuda.synchronize(dev)
            dt = (time.perf_counter() - t0) / 5
            nbytes = buf.numel() * 4
            return {"rccl_world": int(round(float(one.item()))), "allreduce_64MiB_ms": dt * 1e3,
                    "allreduce_busbw_GBps": 2 * (world - 1) / world * nbytes / dt / 1e9,
                    "note": "one all-reduce over all ranks before the span (samplers too); busbw = 2(n-1)/n x "
                            "bytes / time (ranks enter the timed loop unsynchronised: a lower bound)"}
        res, bad = limited_collective(dist, world, body, limit_s)
        if bad:
            ok = False
            try:  # end whatever this rank's thread still has in flight
                grp._get_backend(dev).abort()
            except Exception:
                pass
            rec = {"rccl_world": None, "error": "RCCL self-test failed", "failed_ranks": {str(r): e for r, e in bad.items()},
                   "limit_s": limit_s, "consequence": "the training region (gradient all-reduce over RCCL) is skipped; the "
                                                      "data path has no collective and is measured as usual"}
        else:
            rec = res
    # row of the peer matrix for this rank's GPU: which other ranks' GPUs it can map
    row = []
    for r in range(world):
        other = r % max(n_dev, 1)
        try:
            row.append(True if other == dev_id else bool(torch.cuda.can_device_access_peer(dev_id, other)))
        except Exception:
            row.append(None)
    rows = [None] * world
    dist.all_gather_object(rows, row)
    rec["peer_access"] = {"matrix": rows, "note": "matrix[i][j]: rank i's GPU can map rank j's GPU memory "
                                                  "(hipDeviceCanAccessPeer); ranks sharing a GPU read True"}
    return (rec if rank == 0 else None), ok


# ---- the like-for-like N = 1 point of the N >= 2 pipeline, measured inside the same job ------------------------------
def run_n1_point_child():
    """Child of rank 0 of an N >= 2 job, started BEFORE rank 0 touched the GPU (a process that has initialised the GPU
    never starts another program); waits for one JSON request on stdin -- sent after the job's spans, when the ranks have
    shut their engines down -- then runs the SAME pipeline on ONE GPU: the engine's arch3 (sampler and extractor halves
    of arch5 in one process, background threads, in-process ring; the reference's default single-GPU mode,
    cuda_loops_arch3.cc) on the job's dataset, features in host memory behind the same pre-sample cache, timed by the
    same stamps-and-windows rule.  Prints one JSON line."""
    line = sys.stdin.readline()
    if not line.strip():
        return
    if os.environ.get("FGNN_BENCH_WATCHDOG"):
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["FGNN_BENCH_WATCHDOG"]), exit=True)
    req = json.loads(line)
    for k, v in req["env"].items():
        os.environ[k] = v
    import samgraph.torch as sam
    torch.cuda.set_device(req["dev_id"])
    ctx = "cuda:%d" % req["dev_id"]
    lead, R, K, tail = req["lead"], req["windows"], req["steps"], req["tail"]
    total = lead + R * K + tail
    spe = req["steps_per_epoch"]
    cfg = dict(dataset_path=req["dir"], _arch=sam.kArch3, _sample_type=sam.sample_types[req["sample_type"]],
               batch_size=req["batch_size"], num_epoch=(total + spe - 1) // spe + 1,
               _cache_policy=sam.cache_policies["pre_sample"], presample_epoch=req["presample_epochs"],
               cache_percentage=req["cache_ratio"],
               max_sampling_jobs=10, max_copying_jobs=2, omp_thread_num=8, sampler_ctx=ctx, trainer_ctx=ctx,
               num_fanout=len(req["fanout"]), fanout=req["fanout"], seed=req["seed"])
    sam.config(cfg)
    t0 = time.time()
    sam.init()
    setup = time.time() - t0
    sam.start()
    stamps = []
    with no_gc():
        for _ in range(total):
            key = sam.get_next_batch()
            stamps.append((time.clock_gettime(time.CLOCK_MONOTONIC), key))
    _, wins = read_windows(stamps, lead, R, K)
    win_ms = [(b - a) / K * 1e3 for a, b, _ in wins]
    med = sorted(range(R), key=lambda r: win_ms[r])[(R - 1) // 2]
    edges = sum(sam.get_log_step_value(k // spe, k % spe, sam.kLogL1NumSample) for k in wins[med][2])
    rows = sum(sam.get_log_step_value(k // spe, k % spe, sam.kLogL1FeatureBytes) for k in wins[med][2]) / req["row_bytes"]
    miss = sum(sam.get_log_step_value(k // spe, k % spe, sam.kLogL1MissBytes) for k in wins[med][2]) / req["row_bytes"]
    out = {"value": edges / (wins[med][1] - wins[med][0]), "unit": "edges/s", "n_gpus": 1, "ms_per_step": win_ms[med],
           "windows_ms_per_step": win_ms, "steps": K, "hit_rate": (rows - miss) / max(rows, 1.0), "setup_s": setup,
           "what": "the same pipeline on ONE GPU of this job: arch3 through samgraph.torch / c_lib.so (sampler + extractor "
                   "threads in one process, in-process ring), same dataset, features in host memory behind the same "
                   "pre-sample cache, same windows rule -- the like-for-like N = 1 point of this line (the N = 1 "
                   "bench line itself is config 2's shape: features HBM-resident)"}
    print(json.dumps(out), flush=True)
    sam.shutdown()


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
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
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
