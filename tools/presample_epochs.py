#!/usr/bin/env python3
"""Hit rate of the pre-sample cache policy as a function of RunConfig::presample_epoch (dist/pre_sampler.cc:75-162: the
ranking is by access frequency over that many sampled epochs; the reference's scripts default to 1,
common_config.py:70) -- papers100M-shaped graph, fanout [25,10], batch 8000, cache ratio 0.2 (and 0.1 / 0.3), measured
on two further epochs.  Every epoch reshuffles the train set like the engine does (a permutation per epoch).
usage: python3 tools/presample_epochs.py [--workload papers100M] [--max-epochs 4]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from fgnn_hip import lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="papers100M")
    ap.add_argument("--max-epochs", type=int, default=4)
    ap.add_argument("--graph", default="rmat")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    lib.load()
    w = bench.WORKLOADS[a.workload]
    args = argparse.Namespace(graph=a.graph, workload=a.workload, seed=0x5A4D47)
    indptr, indices, num_edge, desc = bench.gen_graph(args, w, dev)
    train = bench.gen_train_set(args, w, dev)
    bs, N = w["batch_size"], w["num_node"]
    spe = (train.numel() + bs - 1) // bs
    sampler = lib.Sampler(indptr, indices, w["fanout"], bs, sample_type=bench.SAMPLE_TYPES[w["sample_type"]], seed=args.seed)
    bt = sampler.new_batch(0, lib.F32, lib.I64)
    seq = [0]
    g = torch.Generator(device=dev)

    def epoch(e, freq):
        g.manual_seed(1000 + e)
        perm = train[torch.randperm(train.numel(), generator=g, device=dev)].contiguous()
        for step in range(spe):
            seeds = perm[step * bs:min(perm.numel(), (step + 1) * bs)]
            sampler.sample(seeds, e * spe + step, bt, seq=seq[0])
            seq[0] += 1
            lib.presample_count(freq, bt.input_nodes_buffer(), d_num_nodes=bt.d_num_input())
        bt.finish()
        bt.wait()

    test = torch.zeros(N, dtype=torch.int32, device=dev)
    for e in (100, 101):
        epoch(e, test)
    t64 = test.to(torch.int64)
    total = float(t64.sum())
    distinct = int((test > 0).sum())
    print("# %s: %s; %d accesses in two test epochs, %d distinct nodes (%.3f of N)" % (a.workload, desc, int(total), distinct,
                                                                                    distinct / N))
    freq = torch.zeros(N, dtype=torch.int32, device=dev)
    print("presample_epoch  " + "  ".join("hit@%.1f" % r for r in (0.1, 0.2, 0.3)) + "  distinct_seen/N")
    for E in range(1, a.max_epochs + 1):
        epoch(E - 1, freq)
        rank = lib.presample_rank(freq)
        row = []
        for r in (0.1, 0.2, 0.3):
            nc = int(N * r)
            row.append(float(t64[(rank[:nc].to(torch.int64) & 0xFFFFFFFF)].sum()) / total)
        print("%15d  " % E + "  ".join("%7.4f" % x for x in row) + "  %.3f" % (int((freq > 0).sum()) / N), flush=True)
    rank = lib.presample_rank(test)
    row = [float(t64[(rank[:int(N * r)].to(torch.int64) & 0xFFFFFFFF)].sum()) / total for r in (0.1, 0.2, 0.3)]
    print("      hindsight  " + "  ".join("%7.4f" % x for x in row) + "  (ranked by the test epochs themselves: upper bound)")


if __name__ == "__main__":
    main()
