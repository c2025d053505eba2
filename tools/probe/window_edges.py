#!/usr/bin/env python3
"""What a K-step timed window of bench.py's headline pays at its two ends.  The default workload's batch loop
(fgnn_sampler_run_range, 3 streams, 6 buffers) in windows of K batches bracketed like bench.py's (synchronise, clock,
native call, synchronise, clock); every batch carries a device-side stamp of its first sampler kernel's start (100 MHz
wall clock), so the window splits into the batch starts (fill: host-enqueue bound, then the steady interval) and what
follows the last batch's start (its whole latency: three batches are in flight, a batch takes about three intervals).
usage: window_edges.py [K ...]   (default 20 151)"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "fgnn-artifacts_amd"))
import bench  # noqa: E402
from fgnn_hip import lib  # noqa: E402

dev = torch.device("cuda:0")
w = bench.WORKLOADS["papers100M"]
indptr, indices, ne = bench.gen_graph_on_gpu(w["num_node"], w["num_edge"], 42, dev)
feat = bench.gen_features_on_gpu(w["num_node"], w["feat_dim"], dev)
g = torch.Generator(device=dev)
g.manual_seed(7)
label = torch.randint(0, w["num_class"], (w["num_node"],), generator=g, device=dev, dtype=torch.int64)
train = bench.gen_train_set(None, w, dev)  # bench.py's train set (uniform ids, shuffled once)
deg = (indptr[1:].to(torch.int64) - indptr[:-1].to(torch.int64)) & 0xFFFFFFFF
table = torch.full((w["num_node"],), -1, dtype=torch.int32, device=dev)
top = torch.argsort(deg, descending=True)[:int(w["num_node"] * 0.2)]
table[top] = torch.arange(top.numel(), device=dev, dtype=torch.int32)
del deg, top
bs = w["batch_size"]
sampler = lib.Sampler(indptr, indices, w["fanout"], bs, sample_type=lib.KHOP2, seed=0x5A4D47)
batches = [sampler.new_batch(w["feat_dim"], lib.F32, lib.I64) for _ in range(6)]
streams = [torch.cuda.Stream(device=dev) for _ in range(3)]
seq = 0
sampler.run_range(seq, 17, train, bs, batches, streams, cache_table=table, feat=feat, label=label)
seq += 17
for K in [int(a) for a in sys.argv[1:]] or [20, 151]:
    for rep in range(4):
        call = sampler.range_call(seq, K, train, bs, batches, streams, cache_table=table, feat=feat, label=label)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        call.run()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        metas, _, busy = call.results()
        seq += K
        tick = 1e-2  # us per tick
        s0 = min(int(m.t_start) for m in metas)
        starts = sorted((int(m.t_start) - s0) * tick for m in metas)
        gaps = sorted(b - a for a, b in zip(starts, starts[1:]))
        steady = gaps[len(gaps) // 2]
        # the window ends when the LAST batch is done: its start + its latency; K steady intervals would be the time of
        # K batches in the middle of a long run
        print("K %3d rep %d: host bracket %.1f us = %.4f ms/step | median interval between batch starts %.1f us -> "
              "K steady intervals %.1f us | last batch starts at %.1f us: bracket - that = %.1f us (its latency + the "
              "launch and synchronise latencies) | host enqueue %.1f us/batch"
              % (K, rep, el * 1e6, el / K * 1e3, steady, steady * K, starts[-1], el * 1e6 - starts[-1],
                 busy / K * 1e6))
        if rep == 3:
            print("   per batch: input nodes %.0f, cache-split misses %.0f, hits %.0f" % tuple(sum(int(getattr(m, f)) for m in metas) / len(metas) for f in ("num_input", "num_miss", "num_cache")))
        if rep == 3 and K <= 24:
            print("   batch starts (us):", [round(x) for x in starts])
