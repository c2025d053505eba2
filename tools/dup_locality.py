#!/usr/bin/env python3
"""How many of a layer's sampled edges are duplicates, and how many of those could a per-workgroup LDS pre-dedup see?
The sampler kernel gives a workgroup 64 consecutive seeds (<= 64 x fanout edges); a pre-dedup in LDS can only merge
duplicates that fall inside such a tile.  Prints, per layer: edges, share of edges whose neighbour was already known
(duplicates + hits on earlier layers), share that repeats INSIDE its 64-seed tile.
usage: dup_locality.py [papers100M|products|twitter] [batches]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "fgnn-artifacts_amd"))
import bench  # noqa: E402
from fgnn_hip import lib  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "papers100M"
    nb = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    w = bench.WORKLOADS[name]
    dev = torch.device("cuda", 0)
    lib.load()
    indptr, indices, _ = bench.gen_graph_on_gpu(w["num_node"], w["num_edge"], 42, dev)
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    train = torch.randperm(w["num_node"], generator=g, device=dev)[:w["num_train"]].to(torch.int32)
    fan, bs = w["fanout"], w["batch_size"]
    st = "khop2" if w["sample_type"] not in ("khop0", "khop2") else w["sample_type"]
    sampler = lib.Sampler(indptr, indices, fan, bs, sample_type=bench.SAMPLE_TYPES[st], seed=0x5A4D47)
    batch = sampler.new_batch(0, lib.F32, lib.I64)
    tot = [[0, 0, 0] for _ in fan]
    for i in range(nb):
        sampler.sample(train[i * bs:(i + 1) * bs], i, batch)
        batch.finish()
        m = batch.wait()
        for l in range(len(fan)):
            row, col, _, _ = batch.graph(l)
            e = int(m.num_edge[l])
            row, col = row.to(torch.int64), col.to(torch.int64)
            new_nodes = int(m.num_src[l]) - int(m.num_dst[l])
            pairs = (col // 64) * (1 << 32) + row
            tot[l][0] += e
            tot[l][1] += e - new_nodes
            tot[l][2] += e - int(torch.unique(pairs).numel())
    print("%s, %s fanout %s, batch %d, %d batches" % (name, st, fan, bs, nb))
    for l in range(len(fan) - 1, -1, -1):
        e, known, intra = tot[l]
        print("  layer %d: %9d edges/batch, %5.1f %% already known to the table, %5.2f %% repeat inside their 64-seed tile"
              % (l, e // nb, 100.0 * known / max(e, 1), 100.0 * intra / max(e, 1)))


if __name__ == "__main__":
    main()
