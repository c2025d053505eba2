#!/usr/bin/env python3
"""Does the ORDER in which the feature gather visits its rows matter (TLB reach / DRAM locality)?  The same 302 K random
rows of a papers100M-sized table (111 M x 128 f32 = 57 GB), once in random order and once sorted by row id, through
fgnn_gather_rows alone on the GPU."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "fgnn-artifacts_amd"))
from fgnn_hip import lib  # noqa: E402

dev = torch.device("cuda:0")
N, D, U = 111_059_956, 128, 302_000
feat = torch.empty((N, D), dtype=torch.float32, device=dev)
feat[:, 0] = 1.0
g = torch.Generator(device=dev)
g.manual_seed(1)
for name in ("random", "sorted", "random", "sorted"):
    idx = torch.randint(0, N, (U,), generator=g, device=dev, dtype=torch.int64).to(torch.int32)
    if name == "sorted":
        idx = torch.sort(idx.to(torch.int64))[0].to(torch.int32)
    out = torch.empty((U, D), dtype=torch.float32, device=dev)
    for _ in range(3):
        lib.gather_rows(out, feat, idx)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps):
        lib.gather_rows(out, feat, idx)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    print("%s order: %.1f us per launch = %.2f TB/s of 2 x rows x 512 B" % (name, us, 2 * U * D * 4 / us / 1e6))
