#!/usr/bin/env python3
"""Time of one stateless weighted-prefix sampling call (fgnn_sample_weighted_khop_prefix: draws, seed sort, compaction)
at a few seed counts.  usage: stateless_weighted_time.py [path/to/libfgnn_hip.so]  (default: the in-tree build; a
second build can be timed beside it for an A/B of the sort)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "fgnn-artifacts_amd"))

import numpy as np
import torch

from fgnn_hip import lib, synth

if len(sys.argv) > 1:
    lib.use_library(sys.argv[1])


def main():
    num_node = 1 << 22
    indptr, indices = synth.powerlaw_csr(num_node, 40_000_000, seed=3)
    prefix = synth.prob_prefix_table(indptr, indices)
    d = [torch.from_numpy(a.view(np.int32) if a.dtype == np.uint32 else a).cuda() for a in (indptr, indices, prefix)]
    rng = np.random.default_rng(5)
    print(f"# {lib.LIB_PATH}")
    for n in (8000, 22500, 200_000, 1_300_000):
        inp = torch.from_numpy(rng.choice(num_node, size=n, replace=False).astype(np.uint32).view(np.int32)).cuda()
        for _ in range(3):
            lib.sample_weighted_khop_prefix(d[0], d[1], d[2], inp, 5, 1, 7, 1)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        e0.record()
        for r in range(reps):
            lib.sample_weighted_khop_prefix(d[0], d[1], d[2], inp, 5, 1, 7 + r, 1)
        e1.record()
        torch.cuda.synchronize()
        print(f"seeds {n:8d}  fanout 5: {e0.elapsed_time(e1) / reps * 1000:8.1f} us per call (allocations included)")


if __name__ == "__main__":
    main()
