#!/usr/bin/env python3
"""Micro-benchmark of the dominant kernel (gather_rows16_kernel) alone: random 512-byte rows out of a table far
larger than the Infinity Cache, swept over the launch knobs.  Used to pick the defaults; results in profiles/."""
import itertools
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fgnn-artifacts_amd"))
from fgnn_hip import lib  # noqa: E402

dev = torch.device("cuda:0")
rows, dim, n = 32 * 1024 * 1024, 128, 502000      # 16 GiB table, ~one papers100M batch of input nodes
table = torch.empty((rows, dim), dtype=torch.float32, device=dev).uniform_()
idx = torch.randint(0, rows, (n,), device=dev, dtype=torch.int32)
out = torch.empty((n, dim), dtype=torch.float32, device=dev)
bytes_alg = n * (4 + 8 * dim)
print("config,us,GB/s")
for wg, u, nt, nts in itertools.product([3, 4, 6, 8, 16], [2, 4, 8], [0, 1], [0, 1]):
    os.environ.update(FGNN_GATHER_WG_PER_CU=str(wg), FGNN_GATHER_UNROLL=str(u), FGNN_GATHER_NT=str(nt),
                      FGNN_GATHER_NTS=str(nts))
    for _ in range(5):
        lib.gather_rows(out, table, src_index=idx)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        lib.gather_rows(out, table, src_index=idx)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    print(f"wg{wg}_u{u}_nt{nt}_nts{nts},{us:.1f},{bytes_alg / us / 1e3:.0f}")
assert torch.equal(out, table[idx.long()])
