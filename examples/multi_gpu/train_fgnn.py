#!/usr/bin/env python3
"""FGNN training (arch5) through the reference's Python API, laid out like the reference's
example/samgraph/multi_gpu/train_{graphsage,gcn,pinsage}.py: the parent configures and loads the dataset
(`sam.config`, `sam.data_init`) and forks sampler processes (`sample_init` + `sample_once` per step, the pre-sampling
cache policy computed by sampler 0) and trainer processes (`train_init`, `extract_start`, `get_next_batch`,
`get_dgl_blocks[_with_weights]`), linked by the pinned shared-memory queue; trainers synchronise gradients with
torch.distributed (backend "nccl" = RCCL) when there is more than one, or -- `--async`, the reference's
multi_gpu/async/train_graphsage.py -- send them to one model in shared host memory whose optimizer they step.

    # 1 sampler GPU + 1 trainer GPU (BASELINE config 3); --single-gpu puts every worker on cuda:0
    python examples/multi_gpu/train_fgnn.py --model graphsage --dataset-path /tmp/ds/papers --cache-percentage 0.2 \\
        --num-sample-worker 1 --num-train-worker 1
    python examples/multi_gpu/train_fgnn.py --model gcn --sample-type weighted_khop_prefix --fanout 5 10 15 ...
    python examples/multi_gpu/train_fgnn.py --model pinsage ...        # random walks, edge weights = visit counts
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


# run-config entries that belong to the script, not to sam.config()
SCRIPT_KEYS = ("sample_workers", "train_workers", "model", "no_train", "report_acc", "op_by_op", "async_train")


def build_parser():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="graphsage", choices=list(MODELS))
    ap.add_argument("--dataset-path", default="/tmp/fgnn_ds/synth")
    ap.add_argument("--make-dataset", default=None, choices=["products", "small", "learnable"])
    ap.add_argument("--report-acc", type=int, default=0,
                    help="trainer 0: validation accuracy every N of its steps and test accuracy at the end (the "
                         "reference's --report-acc, multi_gpu/train_graphsage.py:65,214-220,344-347,395-397); 0: off")
    ap.add_argument("--op-by-op", action="store_true", help="graphsage: the op-by-op layers instead of the fused ones")
    ap.add_argument("--sample-type", default=None)
    ap.add_argument("--fanout", nargs="+", type=int, default=None)
    ap.add_argument("--batch-size", type=int, default=8000)
    ap.add_argument("--num-epoch", type=int, default=3)
    ap.add_argument("--num-hidden", type=int, default=256)
    ap.add_argument("--lr", type=float, default=0.003)
    ap.add_argument("--dropout", type=float, default=0.5)
    ap.add_argument("--num-sample-worker", type=int, default=1)
    ap.add_argument("--num-train-worker", type=int, default=1)
    ap.add_argument("--single-gpu", action="store_true", help="all workers on cuda:0 (common_config.py:186-191)")
    ap.add_argument("--cache-policy", default="pre_sample", choices=list(sam.cache_policies))
    ap.add_argument("--cache-percentage", type=float, default=0.0)
    ap.add_argument("--no-pipeline", action="store_true")
    ap.add_argument("--async", dest="async_train", action="store_true",
                    help="no gradient all-reduce: every trainer sends its gradients to ONE model in shared host memory, "
                         "steps that model's optimizer and takes the parameters back (multi_gpu/async/train_graphsage.py:"
                         "204-211, 260, 320-326)")
    ap.add_argument("--no-train", action="store_true", help="trainers only consume batches: measures sample + hand-off "
                    "+ extract throughput")
    # PinSAGE (multi_gpu/train_pinsage.py:130-134)
    ap.add_argument("--random-walk-length", type=int, default=3)
    ap.add_argument("--random-walk-restart-prob", type=float, default=0.5)
    ap.add_argument("--num-random-walk", type=int, default=4)
    ap.add_argument("--num-neighbor", type=int, default=5)
    ap.add_argument("--num-layer", type=int, default=3)
    return ap


def parse_args():
    return build_parser().parse_args()


def get_run_config(args):
    rc = dict(dataset_path=args.dataset_path, _arch=sam.kArch5, batch_size=args.batch_size,
              num_epoch=args.num_epoch + 1,  # one warm-up epoch, dropped from the averages (common_config.py:163)
              _cache_policy=sam.cache_policies[args.cache_policy], cache_percentage=args.cache_percentage,
              max_sampling_jobs=10, max_copying_jobs=2, omp_thread_num=max(1, 40 // args.num_train_worker),
              num_sample_worker=args.num_sample_worker, num_train_worker=args.num_train_worker, presample_epoch=1,
              barriered_epoch=0)
    if args.model == "pinsage":
        st = args.sample_type or "random_walk"
        rc.update(random_walk_length=args.random_walk_length, random_walk_restart_prob=args.random_walk_restart_prob,
                  num_random_walk=args.num_random_walk, num_neighbor=args.num_neighbor, num_layer=args.num_layer)
    else:
        st = args.sample_type or ("khop2" if args.model == "graphsage" else "khop0")
        fan = args.fanout or ([25, 10] if args.model == "graphsage" else [5, 10, 15])
        rc.update(num_fanout=len(fan), fanout=fan, num_layer=len(fan))
    rc["_sample_type"] = sam.sample_types[st]
    n_dev = torch.cuda.device_count()
    ns, nt = args.num_sample_worker, args.num_train_worker
    if args.single_gpu or n_dev < ns + nt:
        rc["sample_workers"] = ["cuda:0"] * ns
        rc["train_workers"] = ["cuda:0"] * nt
    else:  # samplers first, trainers after (common_config.py:192-199)
        rc["sample_workers"] = ["cuda:%d" % i for i in range(ns)]
        rc["train_workers"] = ["cuda:%d" % (ns + i) for i in range(nt)]
    rc.update(model=args.model, num_hidden=args.num_hidden, lr=args.lr, dropout=args.dropout,
              pipeline=not args.no_pipeline, no_train=args.no_train, report_acc=args.report_acc, op_by_op=args.op_by_op,
              async_train=args.async_train)
    return rc


def run_sample(worker_id, rc):
    barrier = rc["global_barrier"]
    sam.sample_init(worker_id, rc["sample_workers"][worker_id])
    sam.notify_sampler_ready(barrier)
    num_epoch, num_step = sam.num_epoch(), sam.num_local_step()
    barrier.wait()  # run start
    times, edges = [], 0.0
    for epoch in range(num_epoch):
        barrier.wait()  # epoch start
        tic = time.time()
        for step in range(num_step):
            sam.sample_once()
        times.append(time.time() - tic)
        barrier.wait()  # epoch end
    first = sam.steps_per_epoch() // rc["num_sample_worker"] * worker_id
    for step in range(first, first + num_step):
        edges += sam.get_log_step_value(num_epoch - 1, step, sam.kLogL1NumSample)
    barrier.wait()  # results
    if worker_id == 0:
        sam.report_step_average(num_epoch - 1, first + num_step - 1)
        t = float(np.mean(times[1:])) if len(times) > 1 else times[0]
        print("test_result:sample_time={:.4f}".format(t))
        print("test_result:sampled_edges_per_s={:.4e}".format(edges / max(times[-1], 1e-9)))
    sam.shutdown()


def run_train(worker_id, rc):
    barrier = rc["global_barrier"]
    ctx = rc["train_workers"][worker_id]
    nt = rc["num_train_worker"]
    dev = torch.device(ctx)
    torch.cuda.set_device(dev)
    shared = rc.get("global_cpu_model")  # --async
    if nt > 1 and shared is None:
        torch.distributed.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % rc["dist_port"], rank=worker_id,
                                             world_size=nt, timeout=datetime.timedelta(seconds=600))
    sam.wait_for_sampler_ready(barrier)
    sam.train_init(worker_id, ctx)
    num_layer = rc["num_layer"]
    kw = dict(fused=False) if rc["op_by_op"] and rc["model"] == "graphsage" else {}
    model = MODELS[rc["model"]](sam.feat_dim(), rc["num_hidden"], sam.num_class(), num_layer, rc["dropout"], **kw).to(dev)
    accuracy = None
    if rc["report_acc"] and worker_id == 0 and rc["model"] != "pinsage":
        import train_accuracy
        graph, valid_set, test_set, feat, label = train_accuracy.load_accuracy_data(rc["dataset_path"])
        accuracy = train_accuracy.Accuracy(graph, valid_set, test_set, feat, label, rc["fanout"], rc["batch_size"], dev)
    ddp = nt > 1 and shared is None
    if ddp:
        model = torch.nn.parallel.DistributedDataParallel(model, device_ids=[dev], output_device=dev)
    loss_fcn = nn.CrossEntropyLoss().to(dev)
    if shared is not None:
        model.load_state_dict(shared.state_dict())
        params, cpu_params = list(model.parameters()), list(shared.parameters())
        opt = torch.optim.Adam(params, lr=rc["lr"])  # (only its zero_grad is used: the shared model's optimizer steps)
        cpu_opt = torch.optim.Adam(cpu_params, lr=rc["lr"])
    else:
        opt = torch.optim.Adam(model.parameters(), lr=rc["lr"], fused=True)
    num_epoch, num_step = sam.num_epoch(), sam.steps_per_epoch()
    my_step = num_step // nt + (1 if worker_id < num_step % nt else 0)  # multi_gpu/train_graphsage.py:293-298
    get_blocks = sam.get_dgl_blocks_with_weights if rc["model"] == "pinsage" else sam.get_dgl_blocks
    model.train()
    barrier.wait()  # run start
    totals, copies, trains = [], [], []
    for epoch in range(num_epoch):
        barrier.wait()  # epoch start
        tic = time.time()
        if rc["pipeline"]:
            sam.extract_start(my_step)
        t_train = t_copy = 0.0
        for step in range(my_step):
            t0 = time.time()
            if not rc["pipeline"]:
                sam.sample_once()
            key = sam.get_next_batch()
            blocks, batch_input, batch_label = get_blocks(key, num_layer)
            t1 = time.time()
            if not rc["no_train"]:
                loss = loss_fcn(model(blocks, batch_input), batch_label)
                opt.zero_grad()
                loss.backward()
                if shared is None:
                    opt.step()
                else:
                    with rc["global_lock"], torch.no_grad():
                        for p, c in zip(params, cpu_params):
                            c.grad = p.grad.to("cpu")
                        cpu_opt.step()
                        for p, c in zip(params, cpu_params):
                            p.copy_(c)
            else:
                loss = 0.0
            # the batch's buffers go back to the pool at the next get_next_batch: wait for THIS stream's work only
            # (event_sync of the reference's scripts), not for the extractor thread's copies of the next batches
            torch.cuda.current_stream().synchronize()
            t_copy += t1 - t0
            t_train += time.time() - t1
            if accuracy is not None and (epoch * my_step + step) % rc["report_acc"] == 0:
                tt = time.time()
                acc = accuracy.valid_acc(model.module if ddp else model, dev)
                print("Valid Acc: {:.2f}% | Acc Time: {:.4f} | Total Step: {:d}".format(
                    acc * 100.0, time.time() - tt, epoch * my_step + step))
        totals.append(time.time() - tic)
        copies.append(t_copy)
        trains.append(t_train)
        if worker_id == 0:
            print("Epoch {:03d} | Total {:.4f} s | wait/convert {:.4f} | train {:.4f} | loss {:.4f}".format(
                epoch, totals[-1], t_copy, t_train, float(loss)))
        barrier.wait()  # epoch end
    barrier.wait()  # results
    if accuracy is not None:
        tt = time.time()
        acc = accuracy.test_acc(model.module if ddp else model, dev)
        print("Test Acc: {:.2f}% | Acc Time: {:.4f}".format(acc * 100.0, time.time() - tt))
        print("test_result:test_acc={:.4f}".format(acc))
    if worker_id == 0:
        sl = slice(1, None) if len(totals) > 1 else slice(None)
        print("test_result:pipeline_train_epoch_time={:.4f}".format(float(np.mean(totals[sl]))))
        print("test_result:epoch_time:train_total={:.4f}".format(float(np.mean(trains[sl]))))
        print("test_result:epoch_time:copy_time={:.4f}".format(float(np.mean(copies[sl]))))
    sam.shutdown()


def make_dataset(args):
    """--make-dataset: a synthetic dataset directory in the reference's on-disk format (engine.cc:73-264)"""
    if args.make_dataset:
        from fgnn_hip import synth
        shape = dict(synth.DATASET_SHAPES["products"]) if args.make_dataset == "products" else \
            synth.LEARNABLE_SHAPE if args.make_dataset == "learnable" else \
            dict(num_node=200000, num_edge=4000000, feat_dim=100, num_class=47, num_train=40000)
        root, name = os.path.split(args.dataset_path.rstrip("/"))
        synth.write_dataset(root, name, shape["num_node"], shape["num_edge"], shape["feat_dim"], shape["num_class"],
                            shape["num_train"], shape.get("num_valid", 1000), shape.get("num_test", 1000),
                            learnable=args.make_dataset == "learnable",
                            with_prefix=args.sample_type == "weighted_khop_prefix",
                            with_alias=args.sample_type in ("weighted_khop", "weighted_khop_hash_dedup"))


def main():
    args = parse_args()
    make_dataset(args)
    rc = get_run_config(args)
    ns, nt = rc["num_sample_worker"], rc["num_train_worker"]
    sam.config({k: v for k, v in rc.items() if isinstance(v, (int, float, str, list)) and k not in SCRIPT_KEYS})
    sam.data_init()  # before fork: nothing here touches the GPU
    ctx = mp.get_context("fork")
    if rc["async_train"]:
        # (one CPU thread here: an OpenMP pool created by the parent would not survive the fork)
        torch.set_num_threads(1)
        torch.manual_seed(0)
        kw = dict(fused=False) if rc["op_by_op"] and rc["model"] == "graphsage" else {}
        rc["global_cpu_model"] = MODELS[rc["model"]](sam.feat_dim(), rc["num_hidden"], sam.num_class(), rc["num_layer"],
                                                     rc["dropout"], **kw)
        rc["global_cpu_model"].share_memory()
        rc["global_lock"] = ctx.Lock()
    rc["global_barrier"] = ctx.Barrier(ns + nt)
    rc["dist_port"] = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=run_sample, args=(i, rc)) for i in range(ns)]
    procs += [ctx.Process(target=run_train, args=(i, rc)) for i in range(nt)]
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
