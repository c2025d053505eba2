#!/usr/bin/env python3
"""Builds the R-MAT graph of a dataset shape on the GPU and prints what the hot path is sensitive to: build time, degree
distribution, share of isolated rows.  usage: rmat_probe.py [papers100M|products|twitter|uk-2006-05]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fgnn-artifacts_amd"))
from fgnn_hip import rmat, synth  # noqa: E402

shape = sys.argv[1] if len(sys.argv) > 1 else "papers100M"
w = synth.DATASET_SHAPES[shape]
dev = torch.device("cuda:0")
torch.cuda.synchronize()
t0 = time.time()
indptr, indices, ne = rmat.rmat_csr(w["num_node"], w["num_edge"], 42, dev)
torch.cuda.synchronize()
print("%s: R-MAT CSR N=%d E=%d built in %.1f s, peak torch memory %.1f GB" % (
    shape, w["num_node"], ne, time.time() - t0, torch.cuda.max_memory_allocated() / 1e9), flush=True)
ip = indptr.to(torch.int64) & 0xFFFFFFFF
deg = ip[1:] - ip[:-1]
print("in-degree: max %d, mean %.2f, zero %.3f, >10 %.3f, >25 %.3f" % (
    int(deg.max()), float(deg.double().mean()), float((deg == 0).double().mean()), float((deg > 10).double().mean()),
    float((deg > 25).double().mean())))
tr = rmat.train_set(w["num_node"], w["num_train"], 1, dev).to(torch.int64)
dt = deg[tr]
print("train seeds: zero-degree %.3f, mean min(deg,10) %.2f" % (float((dt == 0).double().mean()),
                                                                 float(dt.clamp(max=10).double().mean())))
