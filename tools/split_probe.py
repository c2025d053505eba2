#!/usr/bin/env python3
"""Times fgnn_get_miss_cache_index (count + scan + split) for 500 K nodes over a 111 M-entry table with a tight grid
(host count) and with the capacity-sized grid the batch driver uses (device count, cap 2.288 M).  Profiling aid."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fgnn-artifacts_amd"))
from fgnn_hip import lib  # noqa: E402

dev = torch.device("cuda:0")
N = 111059956
table = torch.full((N,), -1, dtype=torch.int32, device=dev)
r = torch.randperm(N, device=dev)[: N // 5]
table[r] = torch.arange(N // 5, dtype=torch.int32, device=dev)
U, CAP = 500000, 2288000
nodes = torch.randint(0, N, (CAP,), dtype=torch.int32, device=dev)
d_n = torch.tensor([U], dtype=torch.int32, device=dev)


def t(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


ws = lib.scratch(CAP, dev)
print("tight grid (n=%d host): median %.1f us min %.1f" % ((U,) + t(lambda: lib.get_miss_cache_index(table, nodes[:U], ws=ws))))
print("cap grid (n on device, cap %d): median %.1f us min %.1f" % ((CAP,) + t(
    lambda: lib.get_miss_cache_index(table, nodes, d_num_nodes=d_n, ws=ws))))
print("full cap (n=%d host): median %.1f us min %.1f" % ((CAP,) + t(lambda: lib.get_miss_cache_index(table, nodes, ws=ws))))
