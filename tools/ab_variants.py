#!/usr/bin/env python3
"""A/B of batch-driver switches over ONE generated graph: a sampler per variant (the PROFILING build of the library --
make -C fgnn-artifacts_amd/csrc prof; the shipped one has no switches -- reads its FGNN_* switches when a sampler is
created), timed windows interleaved A B C A B C ... so that box drift hits every variant alike.

  python3 tools/ab_variants.py --variants "base;FGNN_HT_PARTITION=0;FGNN_KHOP_SPLIT_L0=0" \
      [--rounds 5] [--steps 151] [--workload papers100M] [--modes full,sample] [--streams 3]

Prints one line per variant and mode: median / min / max ms per step over the rounds.  The whole path is bench.py's
(`full`: sample + dedup + remap + cache split + feature/label gather; `sample`: without the gather)."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import bench  # noqa: E402
from fgnn_hip import lib  # noqa: E402

lib.use_library(lib.PROF_LIB_PATH)  # the build whose kernels read the FGNN_* switches


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", required=True, help="';'-separated; each 'base' or comma-separated NAME=VALUE")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--steps", type=int, default=151)
    ap.add_argument("--warmup", type=int, default=12)
    ap.add_argument("--workload", default="papers100M")
    ap.add_argument("--modes", default="full,sample")
    ap.add_argument("--streams", type=int, default=3)
    ap.add_argument("--graph", default="rmat")
    ap.add_argument("--sample-type", default=None, help="override the workload's sample type (e.g. khop0)")
    ap.add_argument("--out", default=None)
    ap.add_argument("--timing-variant", action="store_true",
                    help="adds a copy of the first variant whose batches record HIP events around the gather (what "
                         "bench.py does for its roofline line): what the two event records per batch cost")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    lib.load()
    w = bench.WORKLOADS[a.workload]
    if a.sample_type:
        w = dict(w, sample_type=a.sample_type)
    args = argparse.Namespace(graph=a.graph, workload=a.workload, seed=0x5A4D47)
    indptr, indices, num_edge, desc = bench.gen_graph(args, w, dev)
    feat = bench.gen_features_on_gpu(w["num_node"], w["feat_dim"], dev)
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    label = torch.randint(0, w["num_class"], (w["num_node"],), generator=g, device=dev, dtype=torch.int64)
    train = bench.gen_train_set(args, w, dev)
    bs = w["batch_size"]
    spe = (train.numel() + bs - 1) // bs
    table = torch.full((w["num_node"],), -1, dtype=torch.int32, device=dev)
    n_cached = w["num_node"] // 5
    table[torch.randperm(w["num_node"], generator=g, device=dev)[:n_cached]] = torch.arange(n_cached, device=dev,
                                                                                          dtype=torch.int32)
    prefix = bench.gen_prefix_on_gpu(indptr, num_edge, 11, dev) if w["sample_type"] == "weighted_khop_prefix" else None
    streams = [torch.cuda.Stream(device=dev) for _ in range(a.streams)]
    nbuf = 2 * a.streams
    variants = []
    for spec in a.variants.split(";"):
        spec = spec.strip()
        env = {} if spec in ("", "base") else dict(kv.split("=", 1) for kv in spec.split(","))
        saved = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        s = lib.Sampler(indptr, indices, w["fanout"], bs, sample_type=bench.SAMPLE_TYPES[w["sample_type"]],
                        seed=0x5A4D47, prob_prefix=prefix, walk_len=w.get("walk_len", 3), num_walks=w.get("num_walks", 4),
                        restart_prob=w.get("restart_prob", 0.5))
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        variants.append(dict(name=spec or "base", sampler=s, seq=0, env=env,
                             batches=[s.new_batch(w["feat_dim"], lib.F32, lib.I64) for _ in range(nbuf)],
                             ms={m: [] for m in a.modes.split(",")}, edges=0))

    if a.timing_variant:
        v0 = variants[0]
        s2 = lib.Sampler(indptr, indices, w["fanout"], bs, sample_type=bench.SAMPLE_TYPES[w["sample_type"]],
                         seed=0x5A4D47, prob_prefix=prefix, walk_len=w.get("walk_len", 3), num_walks=w.get("num_walks", 4),
                         restart_prob=w.get("restart_prob", 0.5))
        bts = [s2.new_batch(w["feat_dim"], lib.F32, lib.I64) for _ in range(nbuf)]
        for bt in bts:
            bt.enable_timing(True)
        variants.append(dict(name=v0["name"] + " + gather timing events", sampler=s2, seq=0, env=v0["env"], batches=bts,
                             ms={m: [] for m in a.modes.split(",")}, edges=0))

    def region(v, n, mode):
        # switches that the library reads per CALL (the gather's tuning knobs) are in force while this variant runs
        saved = {k: os.environ.get(k) for k in v["env"]}
        os.environ.update(v["env"])
        try:
            return region_(v, n, mode)
        finally:
            for k, val in saved.items():
                if val is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = val

    def region_(v, n, mode):
        # the native batch loop (fgnn_sampler_run_range), like bench.py; the sampler-side stage on two streams
        first = v["seq"]
        full = mode in ("full", "nosplit")  # nosplit: the whole path without the cache-index split (what would merging
        sts = streams if full or len(streams) < 3 else streams[:2]  # the split into another launch buy at most?)
        metas, _, _ = v["sampler"].run_range(first, n, train, bs, v["batches"], sts,
                                             cache_table=None if mode == "nosplit" else table,
                                             feat=feat if full else None, label=label if full else None)
        v["seq"] = first + n
        return sum(int(m.num_edge[l]) for m in metas for l in range(m.num_layers))

    for v in variants:
        region(v, a.warmup, "full")
    torch.cuda.synchronize()
    for r in range(a.rounds):
        for mode in a.modes.split(","):
            for v in variants:
                region(v, 8, mode)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                e = region(v, a.steps, mode)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                v["ms"][mode].append(dt / a.steps * 1e3)
                v["edges"] = e / a.steps
    rows = []
    for v in variants:
        for mode, xs in v["ms"].items():
            rows.append(dict(variant=v["name"], mode=mode, median_ms=float(np.median(xs)), min_ms=min(xs), max_ms=max(xs),
                             windows=[round(x, 4) for x in xs], edges_per_step=v["edges"]))
            print("%-46s %-6s median %.4f  min %.4f  max %.4f ms/step   %s" % (v["name"], mode, np.median(xs), min(xs),
                                                                            max(xs), " ".join("%.4f" % x for x in xs)))
    if a.out:
        with open(a.out, "w") as f:
            json.dump(dict(workload=a.workload, graph=desc, steps=a.steps, rounds=a.rounds, streams=a.streams, rows=rows),
                      f, indent=1)


if __name__ == "__main__":
    main()
