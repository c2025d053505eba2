#!/usr/bin/env python3
"""Where inside the single-pass kernels does the time go?  Runs the bench workload serially with the diagnostic
phase log installed (fgnn_debug_phase_log) and prints, per kernel family, when the workgroups reach each phase
boundary relative to the first workgroup's start (100 MHz wall clock).  Profiling aid."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
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
train = torch.randperm(w["num_node"], generator=g, device=dev)[:w["num_train"]].to(torch.int32)
deg = (indptr[1:].to(torch.int64) - indptr[:-1].to(torch.int64)) & 0xFFFFFFFF
n_cached = int(w["num_node"] * 0.2)
table = torch.full((w["num_node"],), -1, dtype=torch.int32, device=dev)
top = torch.argsort(deg, descending=True)[:n_cached]
table[top] = torch.arange(n_cached, device=dev, dtype=torch.int32)
del top, deg
bs = w["batch_size"]
sampler = lib.Sampler(indptr, indices, w["fanout"], bs, sample_type=lib.KHOP2)
bt = sampler.new_batch(w["feat_dim"], lib.F32, lib.I64)
L = lib.load()
nbytes = L.fgnn_debug_phase_log_bytes()
log = torch.zeros(nbytes // 8, dtype=torch.int64, device=dev)
for i in range(6):
    sampler.run_batch(i, train[i * bs:(i + 1) * bs], i, bt, table, feat, label)
    bt.wait()
torch.cuda.synchronize()
L.fgnn_debug_phase_log(C.c_void_p(log.data_ptr()))
sampler.run_batch(6, train[6 * bs:7 * bs], 6, bt, table, feat, label)
m = bt.wait()
torch.cuda.synchronize()
L.fgnn_debug_phase_log(C.c_void_p(0))
a = log.cpu().numpy().reshape(4, 4096, 8)
names = {0: "sampler (last layer run = layer 0)", 1: "dedup count+assign (layer 0)", 2: "cache split"}
for kind in range(3):
    k = a[kind]
    act = k[:, 0] != 0
    if not act.any():
        continue
    t0 = k[act, 0].min()
    print("%s: %d workgroups" % (names[kind], act.sum()))
    for ph in range(8):
        v = k[act, ph]
        v = v[v != 0]
        if v.size == 0:
            continue
        us = (v - t0) / 100.0
        print("   phase %d reached (us after first start): min %.2f  p10 %.2f  median %.2f  p90 %.2f  max %.2f" % (
            ph, us.min(), np.percentile(us, 10), np.median(us), np.percentile(us, 90), us.max()))


def t_loop(fn, reps=30):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


if os.environ.get("FGNN_PROBE_SPLIT"):
    n_in = int(m["num_input"]) if isinstance(m, dict) else int(m.num_input)
    print("standalone cache split, %d real input nodes: %.1f us/call" % (n_in, t_loop(lambda: bt.cache_index(table))))
    for ab in (1, 2, 4, 8, 3, 7, 15):
        os.environ["FGNN_SPLIT_ABLATE"] = str(ab)
        print("  ablate %2d (1 table load, 2 slot store, 4 output writes, 8 look-back): %.1f us/call" % (
            ab, t_loop(lambda: bt.cache_index(table))))
    del os.environ["FGNN_SPLIT_ABLATE"]
    for gr in (256, 512, 768, 1024):
        os.environ["FGNN_SPLIT_GRID"] = str(gr)
        print("  grid %4d: %.1f us/call" % (gr, t_loop(lambda: bt.cache_index(table))))
    del os.environ["FGNN_SPLIT_GRID"]
    t2 = table.clone()
    print("  same, cloned table: %.1f us/call" % t_loop(lambda: bt.cache_index(t2)))
    inp = bt.input_nodes()
    saved = inp.clone()
    inp.copy_(torch.randint(0, w["num_node"], (inp.numel(),), device=dev, dtype=torch.int32))
    print("  uniform random ids: %.1f us/call" % t_loop(lambda: bt.cache_index(table)))
    inp.copy_(torch.sort(saved.to(torch.int64) & 0xFFFFFFFF)[0].to(torch.int32))
    print("  sorted real ids: %.1f us/call" % t_loop(lambda: bt.cache_index(table)))
    tz = torch.full_like(table, -1)
    inp.copy_(saved)
    print("  real ids, all-miss table: %.1f us/call" % t_loop(lambda: bt.cache_index(tz)))
