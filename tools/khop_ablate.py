#!/usr/bin/env python3
"""Where does khop_sample_kernel spend its time?  Times the fused sampler (layer-0 shape of the papers100M bench:
~60 K seeds, fanout 25) with phases switched off through FGNN_KHOP_ABLATE (bit0 swap simulation, bit1 dedup insert,
bit2 CSR write-back, bit3 neighbour loads).  Profiling aid only."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "fgnn-artifacts_amd"))
import bench  # noqa: E402
from fgnn_hip import lib  # noqa: E402

dev = torch.device("cuda:0")
indptr, indices, ne = bench.gen_graph_on_gpu(111059956, 1615685872, 42, dev)
g = torch.Generator(device=dev)
g.manual_seed(1)
train = torch.randperm(111059956, generator=g, device=dev)[:8000].to(torch.int32)
sampler = lib.Sampler(indptr, indices, [25, 10], 8000, sample_type=lib.KHOP2)
bt = sampler.new_batch()
for mask in (0, 1, 2, 4, 8, 3, 7, 15):
    os.environ["FGNN_KHOP_ABLATE"] = str(mask)
    ts = []
    for it in range(12):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        sampler.sample(train, it, bt)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    print(f"ablate={mask:2d} whole sample chain us: median {sorted(ts)[len(ts)//2]:.1f} min {min(ts):.1f}")
