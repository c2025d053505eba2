#!/usr/bin/env python3
"""GraphSAGE training through the reference's Python API (`import samgraph.torch as sam`), single GPU (arch1),
same call sequence as the reference's example/samgraph/train_graphsage.py: config -> init -> per step
sample_once / get_next_batch / get_dgl_blocks -> forward / backward -> report.

The only difference is the model: DGL has no ROCm wheel in this image, so the mean-aggregator SAGEConv is written
with plain torch ops on the COO blocks the engine returns (row = local id of the sampled neighbour, col = local id
of the seed; the first num_dst source nodes are the seeds themselves).  With DGL installed, get_dgl_blocks returns
DGLBlocks and dgl.nn.SAGEConv works unchanged.

    python examples/train_graphsage.py --dataset-path /tmp/ds/synth --num-epoch 3
    python examples/train_graphsage.py --make-dataset products   # writes a products-shaped synthetic dataset first
    # the reference's train_gcn.py / train_pinsage.py (single process): GCN on khop0 [5,10,15], PinSAGE on random walks
    python examples/train_graphsage.py --model gcn --sample-type khop0 --fanout 5 10 15 ...
    python examples/train_graphsage.py --model pinsage --num-random-walk 4 --num-neighbor 5 ...
"""
import argparse
import os
import sys
import time

import numpy as np
import torch as th
import torch.nn as nn
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fgnn-artifacts_amd"))
sys.path.insert(0, os.path.join(ROOT, "examples"))
import samgraph.torch as sam  # noqa: E402


from models import MODELS, SAGE  # noqa: E402  (examples/models.py: torch-op layers on the engine's COO blocks)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="graphsage", choices=list(MODELS))
    # PinSAGE (example/samgraph/train_pinsage.py: random walks, visit counts as edge weights)
    ap.add_argument("--random-walk-length", type=int, default=3)
    ap.add_argument("--random-walk-restart-prob", type=float, default=0.5)
    ap.add_argument("--num-random-walk", type=int, default=4)
    ap.add_argument("--num-neighbor", type=int, default=5)
    ap.add_argument("--num-layer", type=int, default=3)
    ap.add_argument("--dataset-path", default="/tmp/fgnn_ds/synth")
    ap.add_argument("--make-dataset", default=None, choices=["products", "small", "learnable"],
                    help="write a synthetic dataset of that shape to --dataset-path first (learnable: a small graph whose "
                         "labels follow from features and neighbourhoods, for --report-acc)")
    ap.add_argument("--fanout", nargs="+", type=int, default=[25, 10])
    ap.add_argument("--batch-size", type=int, default=8000)
    ap.add_argument("--num-epoch", type=int, default=3)
    ap.add_argument("--num-hidden", type=int, default=256)
    ap.add_argument("--lr", type=float, default=0.003)
    ap.add_argument("--dropout", type=float, default=0.5)
    ap.add_argument("--sample-type", default="khop2")
    ap.add_argument("--arch", default=None, choices=["arch1", "arch2", "arch3", "arch4"],
                    help="default: arch3 (the reference's default, sampler cuda:0 + trainer cuda:1) with two GPUs, "
                         "arch1 with one")
    ap.add_argument("--cache-percentage", type=float, default=0.0, help="arch2-4: presample cache on the trainer GPU")
    ap.add_argument("--pipeline", action="store_true", help="arch2-4: sam.start() background threads")
    ap.add_argument("--report-acc", type=int, default=0,
                    help="validation accuracy every N steps and test accuracy at the end (the reference's --report-acc, "
                         "multi_gpu/train_graphsage.py:65,344-347,395-397); 0: off")
    ap.add_argument("--op-by-op", action="store_true", help="the op-by-op SAGE layers instead of the fused ones")
    ap.add_argument("--cache-policy", default="pre_sample",
                    help="a key of sam.cache_policies: pre_sample, presample_static, degree, ..., or dynamic_cache "
                         "(arch4 only: the cache is the previous batch's feature tensor; khop0 / khop1 / weighted_khop, "
                         "cache percentage 0)")
    args = ap.parse_args()

    if args.make_dataset:
        from fgnn_hip import synth
        shape = dict(synth.DATASET_SHAPES["products"]) if args.make_dataset == "products" else \
            synth.LEARNABLE_SHAPE if args.make_dataset == "learnable" else \
            dict(num_node=200000, num_edge=4000000, feat_dim=100, num_class=47, num_train=40000)
        root, name = os.path.split(args.dataset_path.rstrip("/"))
        t0 = time.time()
        synth.write_dataset(root, name, shape["num_node"], shape["num_edge"], shape["feat_dim"], shape["num_class"],
                            shape["num_train"], shape.get("num_valid", 1000), shape.get("num_test", 1000),
                            learnable=args.make_dataset == "learnable")
        print("dataset written in {:.1f}s".format(time.time() - t0))

    two = th.cuda.device_count() >= 2
    arch = args.arch or ("arch3" if two else "arch1")
    trainer_ctx = "cuda:1" if (arch in ("arch3", "arch4") and two) else "cuda:0"
    sampler_ctx = "cuda:0" if arch != "arch4" or not two else "cuda:1"
    if arch == "arch4" and two:
        trainer_ctx = "cuda:0"  # builtin_archs['arch4'] (samgraph/common/__init__.py:118-122)
    run_config = dict(dataset_path=args.dataset_path, _arch=sam.builtin_archs[arch]["arch"],
                      _sample_type=sam.sample_types[args.sample_type],
                      batch_size=args.batch_size, num_epoch=args.num_epoch + 1,  # + one warm-up epoch (common_config.py:163)
                      _cache_policy=sam.cache_policies[args.cache_policy], cache_percentage=args.cache_percentage, max_sampling_jobs=10,
                      max_copying_jobs=2, omp_thread_num=8, sampler_ctx=sampler_ctx, trainer_ctx=trainer_ctx)
    if args.model == "pinsage":
        run_config.update(_sample_type=sam.sample_types["random_walk"], random_walk_length=args.random_walk_length,
                          random_walk_restart_prob=args.random_walk_restart_prob, num_random_walk=args.num_random_walk,
                          num_neighbor=args.num_neighbor, num_layer=args.num_layer)
    else:
        run_config.update(num_fanout=len(args.fanout), fanout=args.fanout)
    sam.config(run_config)
    sam.init()
    pipeline = args.pipeline and arch != "arch1"  # arch1 doesn't support pipelining (common_config.py:212-214)
    if pipeline:
        sam.start()
    dev = th.device(trainer_ctx)
    num_layer = args.num_layer if args.model == "pinsage" else len(args.fanout)
    if args.model == "graphsage":
        model = SAGE(sam.feat_dim(), args.num_hidden, sam.num_class(), num_layer, args.dropout,
                     fused=not args.op_by_op).to(dev)
    else:
        model = MODELS[args.model](sam.feat_dim(), args.num_hidden, sam.num_class(), num_layer, args.dropout).to(dev)
    get_blocks = sam.get_dgl_blocks_with_weights if args.model == "pinsage" else sam.get_dgl_blocks
    accuracy = None
    if args.report_acc and args.model != "pinsage":
        import train_accuracy
        graph, valid_set, test_set, feat, label = train_accuracy.load_accuracy_data(args.dataset_path)
        accuracy = train_accuracy.Accuracy(graph, valid_set, test_set, feat, label, args.fanout, args.batch_size,
                                           th.device(sampler_ctx))
    loss_fcn = nn.CrossEntropyLoss()
    opt = th.optim.Adam(model.parameters(), lr=args.lr, fused=True)  # one kernel per step instead of one per tensor and op
    num_epoch, num_step = sam.num_epoch(), sam.steps_per_epoch()
    model.train()
    epoch_total, epoch_sample, epoch_copy, epoch_train, edges = [], [], [], [], 0
    for epoch in range(num_epoch):
        t_epoch = time.time()
        t_train = 0.0
        for step in range(num_step):
            if not pipeline:
                sam.sample_once()
            batch_key = sam.get_next_batch()
            blocks, batch_input, batch_label = get_blocks(batch_key, num_layer)
            t1 = time.time()
            loss = loss_fcn(model(blocks, batch_input), batch_label)
            opt.zero_grad()
            loss.backward()
            opt.step()
            # the batch's buffers go back to the pool at the next get_next_batch: wait for THIS stream's work only
            # (event_sync of the reference's scripts), not for the extractor thread's copies of the next batches
            th.cuda.current_stream().synchronize()
            t_train += time.time() - t1
            if epoch == num_epoch - 1:
                edges += sam.get_log_step_value(epoch, step, sam.kLogL1NumSample)
            if accuracy is not None and (epoch * num_step + step) % args.report_acc == 0:
                tt = time.time()
                acc = accuracy.valid_acc(model, dev)
                print("Valid Acc: {:.2f}% | Acc Time: {:.4f} | Total Step: {:d}".format(
                    acc * 100.0, time.time() - tt, epoch * num_step + step))
        epoch_total.append(time.time() - t_epoch)
        epoch_sample.append(sam.get_log_epoch_value(epoch, sam.kLogEpochSampleTime))
        epoch_copy.append(sam.get_log_epoch_value(epoch, sam.kLogEpochCopyTime))
        epoch_train.append(t_train)
        print("Epoch {:03d} | {:.4f} s | sample {:.4f} | extract {:.4f} | train {:.4f} | loss {:.4f}".format(
            epoch, epoch_total[-1], epoch_sample[-1], epoch_copy[-1], t_train, float(loss)))
    if accuracy is not None:
        tt = time.time()
        acc = accuracy.test_acc(model, dev)
        print("Test Acc: {:.2f}% | Acc Time: {:.4f}".format(acc * 100.0, time.time() - tt))
        print("test_result:test_acc={:.4f}".format(acc))
    sam.report_step_average(num_epoch - 1, num_step - 1)
    # test_result lines in the reference's format (multi_gpu/train_graphsage.py:198-199)
    for k, v in (("epoch_time:total", np.mean(epoch_total[1:])), ("epoch_time:sample_time", np.mean(epoch_sample[1:])),
                 ("epoch_time:copy_time", np.mean(epoch_copy[1:])), ("epoch_time:train_total", np.mean(epoch_train[1:])),
                 ("sampled_edges_per_epoch", edges)):
        print("test_result:{:}={:.4f}".format(k, v))
    sam.shutdown()


if __name__ == "__main__":
    main()
