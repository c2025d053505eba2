#!/usr/bin/env python3
"""Time of the static pre-sampler's counting pass (whole 2-hop neighbourhoods of every batch of one epoch, the engine's
PreSampleStatic loop through the kernel-level C ABI) on a bench-shaped synthetic graph.
usage: presample_static_time.py [papers100M|products|twitter]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "fgnn-artifacts_amd"))
import bench  # noqa: E402
from fgnn_hip import lib  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "papers100M"
    w = bench.WORKLOADS[name]
    dev = torch.device("cuda", 0)
    lib.load()
    indptr, indices, _ = bench.gen_graph_on_gpu(w["num_node"], w["num_edge"], 42, dev)
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    train = torch.randperm(w["num_node"], generator=g, device=dev)[:w["num_train"]].to(torch.int32)
    n, bs, layers = w["num_node"], w["batch_size"], len(w["fanout"])
    stamp = torch.zeros(n, dtype=torch.int32, device=dev)
    freq = torch.zeros(n, dtype=torch.int32, device=dev)
    fronts = [torch.empty(n, dtype=torch.int32, device=dev) for _ in range(2)]
    steps = (train.numel() + bs - 1) // bs
    counts = torch.zeros((steps, layers + 1), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for b in range(steps):
        seeds = train[b * bs:(b + 1) * bs]
        c = counts[b]
        for l in range(layers):
            lib.neighbourhood_expand(indptr, indices, seeds if l == 0 else fronts[(l - 1) & 1], stamp, b + 1, freq,
                                     fronts[l & 1], c[l + 1:l + 2], mark_frontier=(l == 0),
                                     num_frontier=seeds.numel() if l == 0 else 0,
                                     d_num_frontier=None if l == 0 else c[l:l + 1])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    reached = counts[:, 1:].sum(dim=1).float() + bs
    print("%s: %d batches of %d seeds, %d levels: %.3f s (%.2f ms per batch), %.3g nodes reached per batch on average, "
          "%.3g distinct nodes ever reached" % (name, steps, bs, layers, dt, dt / steps * 1e3, float(reached.mean()),
                                                float((freq > 0).sum())))


if __name__ == "__main__":
    main()
