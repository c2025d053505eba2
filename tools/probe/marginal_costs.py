#!/usr/bin/env python3
"""Marginal cost of the stages of the whole path in the three-stream mix (papers100M shape, bench.py's loop): the same
151-batch windows with a stage left out -- no cache split (no table), no feature / label gather, neither."""
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
train = bench.gen_train_set(None, w, dev)
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
K = 151
variants = [("whole path", dict(cache_table=table, feat=feat, label=label)),
            ("no cache split", dict(cache_table=None, feat=feat, label=label)),
            ("no gather", dict(cache_table=table, feat=None, label=None)),
            ("sampling + dedup only", dict(cache_table=None, feat=None, label=None))]
res = {k: [] for k, _ in variants}
sampler.run_range(seq, 17, train, bs, batches, streams, cache_table=table, feat=feat, label=label)
seq += 17
for rep in range(4):
    for name, kw in variants:
        call = sampler.range_call(seq, K, train, bs, batches, streams, **kw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        call.run()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        call.results()
        seq += K
        if rep:
            res[name].append(el / K * 1e6)
for name, _ in variants:
    v = sorted(res[name])
    print("%-24s median %.1f us per batch  (%s)" % (name, v[len(v) // 2], " ".join("%.1f" % x for x in v)))
