#!/usr/bin/env python3
"""The gather kernel alone on COLD rows: every launch gathers a different random row set (8 sets, 2 GB of rows in
total, so nothing is left in the 256 MiB Infinity Cache from the previous use), for two table sizes.  Profiling aid."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fgnn-artifacts_amd"))
from fgnn_hip import lib  # noqa: E402

dev = torch.device("cuda:0")
dim, n = 128, 502000
bytes_alg = n * (4 + 8 * dim)
out = torch.empty((n, dim), dtype=torch.float32, device=dev)
for rows in (32 << 20, 111059956):
    table = torch.empty((rows, dim), dtype=torch.float32, device=dev)
    idxs = [torch.randint(0, rows, (n,), device=dev, dtype=torch.int32) for _ in range(8)]
    for wg, u in ((4, 4), (6, 4), (8, 4), (4, 8), (8, 2), (16, 2)):
        os.environ.update(FGNN_GATHER_WG_PER_CU=str(wg), FGNN_GATHER_UNROLL=str(u))
        for i in range(8):
            lib.gather_rows(out, table, src_index=idxs[i])
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for r in range(48):
            lib.gather_rows(out, table, src_index=idxs[r % 8])
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 48 * 1e3
        print(f"table {rows * dim * 4 / 2**30:.0f} GiB wg{wg} u{u}: {us:.1f} us  {bytes_alg / us / 1e3:.0f} GB/s")
    del table, idxs
