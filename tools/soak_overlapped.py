#!/usr/bin/env python3
"""Soak of the OVERLAPPED batch loop against the oracle: fgnn_sampler_run_range (the loop bench.py times: whole batches
rotating over three streams, sample -> cache split -> feature / label gather) on the products-shaped graph, many rounds;
after every round each batch of the round -- blocks, node list, cache index arrays, gathered rows and labels -- is
compared bit-exactly with the oracle's replay of the same batch sequence (khop2: the CSR mutation carries from batch to
batch and round to round on both sides, and the mutated CSR is compared at the end).  The parity tests compare batches
run one at a time; this is the same comparison with the batches in flight together, at a realistic size.

usage: soak_overlapped.py [--kind khop2|khop0|khop1|weighted_khop_prefix|weighted_khop|random_walk] [--fanout 10,5,5] [--rounds 12] [--per-round 48] [--streams 3]
       [--help-after 0]
GPU box; the oracle (CPU) is most of the run time (~0.1-0.2 s per batch)."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "fgnn-artifacts_amd"), os.path.join(ROOT, "oracle"), ROOT):
    sys.path.insert(0, p)

import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kind", default="khop2", choices=["khop2", "khop0", "khop1", "weighted_khop_prefix", "weighted_khop",
                                                         "random_walk"])
    ap.add_argument("--fanout", default="10,5,5")
    ap.add_argument("--rounds", type=int, default=12)
    ap.add_argument("--per-round", type=int, default=48)
    ap.add_argument("--streams", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8000)
    ap.add_argument("--seed", type=lambda s: int(s, 0), default=0x5A4D47)
    ap.add_argument("--help-after", type=int, default=-1,
                    help="polls before a waiting workgroup of the single-pass kernels computes its predecessors' "
                         "share itself (fgnn_debug_set_scan_help_after; 0 = always: the fallback path under load)")
    run(ap.parse_args())


def run(args):
    import bench
    import oracle_py as oracle
    from fgnn_hip import lib
    oracle.build()
    lib.load()
    if args.help_after >= 0:
        lib.load().fgnn_debug_set_scan_help_after(args.help_after)
    dev = torch.device("cuda:0")
    fanout = [int(x) for x in args.fanout.split(",")]
    w = bench.WORKLOADS["products"]
    N, D, B = w["num_node"], w["feat_dim"], args.batch
    indptr, indices, _ = bench.gen_graph_on_gpu(N, w["num_edge"], 42, dev)
    feat = bench.gen_features_on_gpu(N, D, dev)
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    label = torch.randint(0, w["num_class"], (N,), generator=g, device=dev, dtype=torch.int64)
    train = torch.randperm(N, generator=g, device=dev)[:w["num_train"]].to(torch.int32)
    rank = torch.randperm(N, generator=g, device=dev).to(torch.int32)
    table = lib.cache_table_build(rank, N // 5, N)
    h_indptr = indptr.cpu().numpy().view(np.uint32)
    h_indices = indices.cpu().numpy().view(np.uint32).copy()
    h_table = table.cpu().numpy().view(np.uint32)
    h_train = train.cpu().numpy().view(np.uint32)
    h_label = label.cpu().numpy()
    st, ost = dict(khop2=(lib.KHOP2, oracle.KHOP2), khop0=(lib.KHOP0, oracle.KHOP0), khop1=(lib.KHOP1, oracle.KHOP1),
                   weighted_khop_prefix=(lib.WEIGHTED_KHOP_PREFIX, oracle.WEIGHTED_KHOP_PREFIX),
                   weighted_khop=(lib.WEIGHTED_KHOP, oracle.WEIGHTED_KHOP),
                   random_walk=(lib.RANDOM_WALK, oracle.RANDOM_WALK))[args.kind]
    d_indices = indices.clone()
    kw, okw = {}, {}
    to_dev = lambda a: torch.from_numpy(a.view(np.int32) if a.dtype == np.uint32 else a).to(dev)  # noqa: E731
    if args.kind == "weighted_khop_prefix":
        from fgnn_hip import synth
        prefix = synth.prob_prefix_table(h_indptr, h_indices)
        kw, okw = dict(prob_prefix=to_dev(prefix)), dict(prob_prefix=prefix)
    elif args.kind == "weighted_khop":
        from fgnn_hip import synth
        prob, alias = synth.alias_tables(h_indptr, h_indices)
        kw, okw = dict(prob_table=to_dev(prob), alias_table=to_dev(alias)), dict(prob_prefix=prob, alias_table=alias)
    elif args.kind == "random_walk":  # PinSAGE's sampler as BASELINE config 5 runs it: 25 walks of 3 steps, top-K = fanout
        assert len(set(fanout)) == 1
        kw = dict(walk_len=3, num_walks=25, restart_prob=0.5)
        okw = dict(walk_len=3, num_walks=25, num_neighbor=fanout[0], restart_prob=0.5)
    sampler = lib.Sampler(indptr, d_indices, fanout, B, sample_type=st, seed=args.seed, **kw)
    batches = [sampler.new_batch(D, lib.F32, lib.I64) for _ in range(args.per_round)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(args.streams)]
    rng = oracle.make_rng(oracle.RNG_PHILOX, args.seed)
    oht = oracle.HashTable(N, oracle.predict_num_nodes(B, fanout))
    steps = (len(h_train) + B - 1) // B
    L = len(fanout)
    t_start = time.time()
    checked = edges = 0
    for r in range(args.rounds):
        first = r * args.per_round
        torch.cuda.synchronize()
        metas, _, _ = sampler.run_range(first, args.per_round, train, B, batches, streams, cache_table=table, feat=feat,
                                        label=label)
        torch.cuda.synchronize()
        for i in range(args.per_round):
            seq = first + i
            step = seq % steps
            seeds = h_train[step * B:min((step + 1) * B, len(h_train))]
            bt, m = batches[seq % len(batches)], metas[i]
            want = oracle.do_sample(h_indptr, h_indices, seeds, fanout, ost, rng, step, oht, **okw)
            what = "%s round %d batch %d (seq %d, step %d)" % (args.kind, r, i, seq, step)
            assert m.overflow == 0 and m.num_output == len(seeds) and m.key == step, what
            nodes = bt.input_nodes().cpu().numpy().view(np.uint32)
            np.testing.assert_array_equal(nodes, want["input_nodes"], err_msg=what)
            for l in range(L):
                row, col, nsrc, ndst = bt.graph(l)
                gr = want["graphs"][l]
                assert (nsrc, ndst, int(m.num_edge[l])) == (gr["num_src"], gr["num_dst"], gr["num_edge"]), (what, l)
                np.testing.assert_array_equal(row.cpu().numpy().view(np.uint32), gr["row"], err_msg="%s layer %d" % (what, l))
                np.testing.assert_array_equal(col.cpu().numpy().view(np.uint32), gr["col"], err_msg="%s layer %d" % (what, l))
                if args.kind == "random_walk":
                    np.testing.assert_array_equal(bt.graph_data(l).cpu().numpy().view(np.uint32), gr["data"], err_msg=what)
                edges += gr["num_edge"]
            for got, wv in zip(bt.cache_index_arrays(), oracle.get_miss_cache_index(h_table, nodes)):
                np.testing.assert_array_equal(got.cpu().numpy().view(np.uint32), wv, err_msg=what)
            # rows and labels: the oracle's extraction is a row copy -- compared on the GPU against an index_select
            idx = torch.from_numpy(nodes.astype(np.int64)).to(dev)
            assert torch.equal(bt.feat().view(torch.int32), feat.index_select(0, idx).view(torch.int32)), what
            np.testing.assert_array_equal(bt.label().cpu().numpy(), h_label[seeds], err_msg=what)
            checked += 1
        print("round %d: %d batches identical so far (%d sampled edges), %.0f s" % (r, checked, edges,
                                                                                   time.time() - t_start), flush=True)
    np.testing.assert_array_equal(d_indices.cpu().numpy().view(np.uint32), h_indices)
    print("soak ok: %s fanout %s, %d batches over %d streams identical to the oracle's replay, CSR identical at the end"
          % (args.kind, fanout, checked, args.streams))


if __name__ == "__main__":
    main()
