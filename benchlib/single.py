"""benchlib.single -- bench.py at N = 1: one GPU samples AND extracts through the kernel-level C ABI; the headline windows,
the sampler-side stage, config 3's extract leg, the training leg."""
import json
import os
import sys
import time

from .common import (  # noqa: F401
    HBM_PEAK_GBS, HOST_LINK_GBS, HostTable, ROOT, SAMPLE_TYPES, WORKLOADS, algorithmic_bytes,
    gen_alias_on_gpu, gen_features_on_gpu, gen_graph, gen_prefix_on_gpu, gen_train_set,
    gpu_numa_node, lib, no_gc, np, numa_nodes_with_memory, pages_by_numa_node, pmc_requests,
    pmc_traffic, torch)
from .cpu import (  # noqa: F401
    cpu_baseline, cpu_baseline_generic, cpu_baseline_products)
from .pipeline import (  # noqa: F401
    train_region_batches)


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
    gather_clock_ms = []  # the same launches by their own clock stamps (native loop only)
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
            gather_clock_ms.extend(t[1] if mode[0] == "full" else -1.0 for t in tm)
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
        gather_clock_ms.clear()
        for t in range(NT):
            host_busy[t] = 0.0
        el = timed_region(seq0 + r * args.steps, seq0 + (r + 1) * args.steps)
        assert len(metas) == args.steps, (len(metas), args.steps)
        windows.append(dict(elapsed=el, metas=list(metas), gather_ms=list(gather_ms), gather_clock_ms=list(gather_clock_ms),
                            host_enqueue_ms=sum(host_busy) / args.steps * 1e3))
    order = sorted(range(R), key=lambda r: windows[r]["elapsed"])
    med = windows[order[(R - 1) // 2]]  # the median window (the slower of the middle two for an even R)
    elapsed = med["elapsed"]
    host_enqueue_ms = med["host_enqueue_ms"]  # of that timed window only
    metas[:] = med["metas"]
    gather_ms[:] = med["gather_ms"]
    gather_clock_t = [x for x in med["gather_clock_ms"] if x >= 0]
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
                     # the same launches by the kernel's OWN clock (first workgroup's start .. last one's end, 100 MHz
                     # words posted to pinned memory): what a kernel trace reports as the duration -- the HIP-event
                     # bracket above also holds the launch gap and the wait for wave slots behind other batches' kernels
                     "kernel_clock": ({"avg_kernel_ms": float(np.mean(gather_clock_t)),
                                       "achieved": gather_feat_bytes / len(metas) / (float(np.mean(gather_clock_t)) * 1e-3) / 1e9,
                                       "frac": gather_feat_bytes / len(metas) / (float(np.mean(gather_clock_t)) * 1e-3) / 1e9
                                       / HBM_PEAK_GBS, "launches": len(gather_clock_t)} if gather_clock_t else None),
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

    # The host never waits for a training step: the step of batch j is LAUNCHED (one graph replay) and the loop goes on to
    # wait for batch j + 1's summary, enqueue batch j + 2 and launch step j + 1 while step j still runs.  What a batch
    # buffer's next sampling must wait for -- the step that read the buffer, two batches ago -- is a GPU-side event wait
    # on the sampling stream.  (Until round 6 the loop synchronised after every step like the reference's scripts do.  The
    # leg did not get faster -- 0.505-0.526 -> 0.500 ms per step: the next batch's sampling and extraction used to run in
    # the shadow of the host's 0.08 ms between two steps, now they run beside the step and both stretch; the GPU is bound
    # by the SUM of the step's 0.32 ms and the batch's 0.10 ms -- but the leg no longer depends on the host's speed.)
    step_done = [torch.cuda.Event(), torch.cuda.Event()]
    step_recorded = [False, False]

    def enqueue(i):
        step, seeds = seeds_of(i)
        if step_recorded[i % 2]:
            st.wait_event(step_done[i % 2])  # the step that read this buffer (batch i - 2) has finished
        sampler.run_batch(i, seeds, step, bufs[i % 2], table, feat, label, stream=st)

    phases = {"wait_for_batch": 0.0, "enqueue_next_batch": 0.0, "launch_step": 0.0, "wait_for_last_step": 0.0}

    def region(first, n):
        torch.cuda.synchronize()
        for k in phases:
            phases[k] = 0.0
        step_recorded[0] = step_recorded[1] = False  # (everything before the synchronise above has finished)
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
            step_done[(first + j) % 2].record()
            step_recorded[(first + j) % 2] = True
            phases["launch_step"] += time.perf_counter() - tc
        td = time.perf_counter()
        torch.cuda.synchronize()
        phases["wait_for_last_step"] += time.perf_counter() - td
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
            "loop": "no host wait per step: step j is launched, then batch j + 1's summary is awaited, batch j + 2 enqueued "
                    "(its buffer's last reader -- step j -- by a GPU-side event) and step j + 1 launched while step j runs",
            "what": "sample + extract of batch k+1 on a side stream under the GraphSAGE step of batch k (examples/models.py: "
                    f"{L} fused SAGEConv layers, hidden 256, fp32, fused Adam; aggregation by fgnn_block_aggregate), one "
                    "GPU"}, next_seq
