#!/usr/bin/env python3
"""The reference's SGNN baselines through the same Python API, laid out like example/samgraph/sgnn/train_*.py (arch6)
and example/samgraph/sgnn_dgl/train_*.py (arch7): every worker process owns one GPU and does everything on it --
sample, extract (arch6: `sample_init` + `train_init` in the worker, features gathered by the engine; arch7: a
sample-only engine per worker via `config` + `init`, features through `load_subtensor`), train -- over an equal share
of the shuffled train set (DistAlignedShuffler); workers synchronise gradients with torch.distributed.

    python examples/sgnn/train_sgnn.py --arch arch6 --num-worker 8 --dataset-path /tmp/ds/papers --cache-percentage 0.1
    python examples/sgnn/train_sgnn.py --arch arch7 --num-worker 2 --single-gpu --make-dataset small
"""
import argparse
import datetime
import os
import sys
import time

import numpy as np
import torch
import torch.multiprocessing as mp
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "fgnn-artifacts_amd"))
sys.path.insert(0, os.path.join(ROOT, "examples"))
import samgraph.torch as sam  # noqa: E402
from models import MODELS  # noqa: E402


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--arch", default="arch6", choices=["arch6", "arch7"])
    ap.add_argument("--model", default="graphsage", choices=list(MODELS))
    ap.add_argument("--dataset-path", default="/tmp/fgnn_ds/synth")
    ap.add_argument("--make-dataset", default=None, choices=["products", "small"])
    ap.add_argument("--sample-type", default=None)
    ap.add_argument("--fanout", nargs="+", type=int, default=None)
    ap.add_argument("--batch-size", type=int, default=8000)
    ap.add_argument("--num-epoch", type=int, default=3)
    ap.add_argument("--num-hidden", type=int, default=256)
    ap.add_argument("--lr", type=float, default=0.003)
    ap.add_argument("--dropout", type=float, default=0.5)
    ap.add_argument("--num-worker", type=int, default=1)
    ap.add_argument("--single-gpu", action="store_true", help="all workers on cuda:0 (common_config.py:186-191)")
    ap.add_argument("--cache-policy", default="pre_sample", choices=list(sam.cache_policies))
    ap.add_argument("--cache-percentage", type=float, default=0.0)
    ap.add_argument("--random-walk-length", type=int, default=3)
    ap.add_argument("--random-walk-restart-prob", type=float, default=0.5)
    ap.add_argument("--num-random-walk", type=int, default=4)
    ap.add_argument("--num-neighbor", type=int, default=5)
    ap.add_argument("--num-layer", type=int, default=3)
    return ap.parse_args()


def get_run_config(args):
    nw = args.num_worker
    rc = dict(dataset_path=args.dataset_path, _arch=sam.builtin_archs[args.arch]["arch"], batch_size=args.batch_size,
              num_epoch=args.num_epoch + 1,  # one warm-up epoch (common_config.py:163)
              _cache_policy=sam.cache_policies[args.cache_policy],
              cache_percentage=args.cache_percentage if args.arch == "arch6" else 0.0,  # arch7 has no cache
              max_sampling_jobs=10, max_copying_jobs=2, omp_thread_num=max(1, 40 // nw), num_worker=nw,
              presample_epoch=1, barriered_epoch=0)
    if args.model == "pinsage":
        st = args.sample_type or "random_walk"
        rc.update(random_walk_length=args.random_walk_length, random_walk_restart_prob=args.random_walk_restart_prob,
                  num_random_walk=args.num_random_walk, num_neighbor=args.num_neighbor, num_layer=args.num_layer)
    else:
        st = args.sample_type or ("khop2" if args.model == "graphsage" else "khop0")
        fan = args.fanout or ([25, 10] if args.model == "graphsage" else [5, 10, 15])
        rc.update(num_fanout=len(fan), fanout=fan, num_layer=len(fan))
    rc["_sample_type"] = sam.sample_types[st]
    shared = args.single_gpu or torch.cuda.device_count() < nw
    rc["workers"] = ["cuda:0"] * nw if shared else ["cuda:%d" % i for i in range(nw)]
    # RCCL refuses two ranks on one device: the one-GPU layout synchronises gradients over gloo
    rc["dist_backend"] = "gloo" if shared else "nccl"
    rc.update(model=args.model, num_hidden=args.num_hidden, lr=args.lr, dropout=args.dropout, arch_name=args.arch)
    return rc


def engine_config(rc):
    return {k: v for k, v in rc.items() if isinstance(v, (int, float, str, list)) and k not in
            ("workers", "model", "arch_name", "dist_backend")}


def run_worker(worker_id, rc):
    barrier = rc["global_barrier"]
    nw = rc["num_worker"]
    ctx = rc["workers"][worker_id]
    dev = torch.device(ctx)
    torch.cuda.set_device(dev)
    arch7 = rc["arch_name"] == "arch7"
    if arch7:  # sgnn_dgl/train_graphsage.py:95-100: the engine is configured per worker
        cfg = engine_config(rc)
        cfg.update(worker_id=worker_id, sampler_ctx=ctx, trainer_ctx=ctx)
        sam.config(cfg)
        sam.init()
        feat, label = sam.get_dataset_feat(), sam.get_dataset_label()
    else:      # sgnn/train_graphsage.py:117-118
        sam.sample_init(worker_id, ctx)
        sam.train_init(worker_id, ctx)
    if nw > 1:
        torch.distributed.init_process_group(rc["dist_backend"], init_method="tcp://127.0.0.1:%d" % rc["dist_port"],
                                             rank=worker_id, world_size=nw,
                                             timeout=datetime.timedelta(seconds=600))
    num_layer = rc["num_layer"]
    model = MODELS[rc["model"]](sam.feat_dim(), rc["num_hidden"], sam.num_class(), num_layer, rc["dropout"]).to(dev)
    if nw > 1:
        model = torch.nn.parallel.DistributedDataParallel(model, device_ids=[dev], output_device=dev)
    loss_fcn = nn.CrossEntropyLoss().to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=rc["lr"], fused=True)
    num_epoch, num_step = sam.num_epoch(), sam.num_local_step()
    get_blocks = sam.get_dgl_blocks_with_weights if rc["model"] == "pinsage" else sam.get_dgl_blocks
    model.train()
    barrier.wait()  # run start
    totals, samples, copies, trains, edges = [], [], [], [], 0.0
    for epoch in range(num_epoch):
        barrier.wait()  # epoch start
        tic = time.time()
        t_sample = t_copy = t_train = 0.0
        for step in range(num_step):
            t0 = time.time()
            sam.sample_once()
            key = sam.get_next_batch()
            t1 = time.time()
            if arch7:
                batch_input, batch_label = sam.load_subtensor(key, feat, label, dev)
                blocks, _, _ = get_blocks(key, num_layer, with_feat=False)
            else:
                blocks, batch_input, batch_label = get_blocks(key, num_layer)
            t2 = time.time()
            loss = loss_fcn(model(blocks, batch_input), batch_label)
            opt.zero_grad()
            loss.backward()
            opt.step()
            # the batch's buffers go back to the pool at the next get_next_batch: wait for THIS stream's work only
            # (event_sync of the reference's scripts), not for the extractor thread's copies of the next batches
            torch.cuda.current_stream().synchronize()
            t_sample += t1 - t0
            t_copy += t2 - t1
            t_train += time.time() - t2
        if nw > 1:
            torch.distributed.barrier()
        totals.append(time.time() - tic)
        samples.append(t_sample)
        copies.append(t_copy)
        trains.append(t_train)
        if worker_id == 0:
            print("Epoch {:03d} | Total {:.4f} s | sample+extract {:.4f} | convert/copy {:.4f} | train {:.4f} | "
                  "loss {:.4f}".format(epoch, totals[-1], t_sample, t_copy, t_train, float(loss)))
        barrier.wait()  # epoch end
    first = sam.num_local_step() * worker_id  # this worker's global steps (dist_shuffler_aligned.h:41)
    for step in range(first, first + num_step):
        edges += sam.get_log_step_value(num_epoch - 1, step, sam.kLogL1NumSample)
    barrier.wait()  # results
    if worker_id == 0:
        sl = slice(1, None) if len(totals) > 1 else slice(None)
        print("test_result:epoch_time:total={:.4f}".format(float(np.mean(totals[sl]))))
        print("test_result:epoch_time:sample_total={:.4f}".format(float(np.mean(samples[sl]))))
        print("test_result:epoch_time:copy_time={:.4f}".format(float(np.mean(copies[sl]))))
        print("test_result:epoch_time:train_total={:.4f}".format(float(np.mean(trains[sl]))))
        print("test_result:sampled_edges_per_epoch_per_worker={:.0f}".format(edges))
    sam.shutdown()


def main():
    args = parse_args()
    if args.make_dataset:
        from fgnn_hip import synth
        shape = dict(synth.DATASET_SHAPES["products"]) if args.make_dataset == "products" else \
            dict(num_node=200000, num_edge=4000000, feat_dim=100, num_class=47, num_train=40000)
        root, name = os.path.split(args.dataset_path.rstrip("/"))
        synth.write_dataset(root, name, shape["num_node"], shape["num_edge"], shape["feat_dim"], shape["num_class"],
                            shape["num_train"], 1000, 1000,
                            with_prefix=args.sample_type == "weighted_khop_prefix",
                            with_alias=args.sample_type in ("weighted_khop", "weighted_khop_hash_dedup"))
    rc = get_run_config(args)
    nw = rc["num_worker"]
    if args.arch == "arch6":
        sam.config(engine_config(rc))
        sam.data_init()  # before fork: nothing here touches the GPU
    ctx = mp.get_context("fork")
    rc["global_barrier"] = ctx.Barrier(nw)
    rc["dist_port"] = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=run_worker, args=(i, rc)) for i in range(nw)]
    for p in procs:
        p.start()
    ret = sam.wait_one_child()
    if ret != 0:
        for p in procs:
            p.kill()
    for p in procs:
        p.join()
    if ret != 0:
        sys.exit(1)


if __name__ == "__main__":
    main()
