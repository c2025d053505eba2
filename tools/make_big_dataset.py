#!/usr/bin/env python3
"""Writes a papers100M-/products-shaped synthetic dataset in the engine's on-disk layout WITHOUT feat.bin (run the
engine with SAMGRAPH_EMPTY_FEAT=k, like the reference's papers100M_empty): the CSR comes from bench.py's GPU generator,
so a 1.6 G-edge graph is written in seconds.
usage: make_big_dataset.py <dir> [papers100M|products|twitter|uk-2006-05] [--prefix]   (--prefix: also
prob_prefix_table.bin for the weighted_khop_prefix sampler)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "fgnn-artifacts_amd"))
import bench  # noqa: E402

out, shape = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "papers100M")
w = bench.WORKLOADS[shape]
os.makedirs(out, exist_ok=True)
dev = torch.device("cuda:0")
indptr, indices, ne = bench.gen_graph_on_gpu(w["num_node"], w["num_edge"], 42, dev)
indptr.cpu().numpy().view(np.uint32).tofile(os.path.join(out, "indptr.bin"))
chunk = 1 << 28
with open(os.path.join(out, "indices.bin"), "wb") as f:
    for a in range(0, ne, chunk):
        f.write(indices[a:a + chunk].cpu().numpy().view(np.uint32).tobytes())
g = torch.Generator(device=dev)
g.manual_seed(7)
label = torch.randint(0, w["num_class"], (w["num_node"],), generator=g, device=dev, dtype=torch.int64)
label.cpu().numpy().astype(np.uint64).tofile(os.path.join(out, "label.bin"))
deg = (indptr[1:].to(torch.int64) - indptr[:-1].to(torch.int64)) & 0xFFFFFFFF
cand = torch.nonzero(deg > 0).flatten()
perm = cand[torch.randperm(cand.numel(), generator=g, device=dev)]
n_tr = w["num_train"]
sets = {"train": perm[:n_tr], "valid": perm[n_tr:n_tr + 1000], "test": perm[n_tr + 1000:n_tr + 2000]}
for k, v in sets.items():
    v.to(torch.int32).cpu().numpy().view(np.uint32).tofile(os.path.join(out, k + "_set.bin"))
if "--prefix" in sys.argv:
    pre = bench.gen_prefix_on_gpu(indptr, ne, 11, dev)
    with open(os.path.join(out, "prob_prefix_table.bin"), "wb") as f:
        for a in range(0, ne, chunk):
            f.write(pre[a:a + chunk].cpu().numpy().tobytes())
    del pre
with open(os.path.join(out, "meta.txt"), "w") as f:
    f.write(f"NUM_NODE {w['num_node']}\nNUM_EDGE {ne}\nFEAT_DIM {w['feat_dim']}\nNUM_CLASS {w['num_class']}\n"
            f"NUM_TRAIN_SET {n_tr}\nNUM_VALID_SET 1000\nNUM_TEST_SET 1000\n")
print("wrote", out, "edges", ne)
