#!/usr/bin/env python3
"""Phase timestamps of the FIRST-layer sampler launch (8000 seeds, fanout 10): a single-layer sampler so that the
diagnostic log is not overwritten by a later layer.  Profiling aid."""
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
g = torch.Generator(device=dev)
g.manual_seed(7)
train = torch.randperm(w["num_node"], generator=g, device=dev)[:w["num_train"]].to(torch.int32)
bs = 8000
sampler = lib.Sampler(indptr, indices, [10], bs, sample_type=lib.KHOP2)
bt = sampler.new_batch()
L = lib.load()
log = torch.zeros(L.fgnn_debug_phase_log_bytes() // 8, dtype=torch.int64, device=dev)
for i in range(6):
    sampler.sample(train[i * bs:(i + 1) * bs], i, bt)
    bt.finish()
    bt.wait()
L.fgnn_debug_phase_log(C.c_void_p(log.data_ptr()))
sampler.sample(train[6 * bs:7 * bs], 6, bt)
bt.finish()
bt.wait()
torch.cuda.synchronize()
L.fgnn_debug_phase_log(C.c_void_p(0))
a = log.cpu().numpy().reshape(4, 4096, 8)
k = a[0]
act = k[:, 0] != 0
t0 = k[act, 0].min()
print("first-layer sampler: %d workgroups" % act.sum())
for ph in range(5):
    v = k[act, ph]
    v = v[v != 0]
    if v.size:
        us = (v - t0) / 100.0
        print("   phase %d: min %.2f median %.2f max %.2f" % (ph, us.min(), np.median(us), us.max()))
