"""benchlib.cpu -- the CPU baselines of the bench line: the reference's own CPU sources (oracle/_ref) or the oracle, timed
on the host (the only users of oracle/ outside tests/)."""
import os
import sys
import time

from .common import (  # noqa: F401
    ROOT, np, synth, torch)


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
