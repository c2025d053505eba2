#!/usr/bin/env python3
"""FGNN training with the DYNAMIC SWITCHER (BASELINE config 5) through the reference's Python API, laid out like the
reference's example/samgraph/balance_switcher/train_{pinsage,graphsage,gcn}.py:

  * every sampler GPU carries TWO processes: the sampler (`sample_init`, `sample_once` per step) and its switcher
    (`switch_init(worker_id, ctx, switch_cache_percentage)`), which sleeps until the sampler has produced its share of
    the epoch and then trains on that GPU from the same queue as the trainers (balance_switcher/train_pinsage.py:305-309,
    354-366; dist_engine.cc:425-431);
  * `have_switcher` makes the samplers ship input nodes instead of (miss, hit) index pairs: every consumer splits them
    against ITS OWN cache (task_queue.cc:93-95) -- a switcher's cache may be smaller than a trainer's, its GPU also holds
    the graph;
  * one permit per batch of the epoch in a shared semaphore, released by the samplers at the epoch's start; trainers and
    switchers take a permit, dequeue (`sample_once` + `get_next_batch`) and train until no permit is left
    (train_pinsage.py:219-220, 363-366);
  * the consumers are asynchronous -- a switcher joins in the middle of an epoch -- so gradients are not all-reduced:
    every step's parameter DELTA is folded into one model in shared host memory under a lock and the result taken back
    (train_pinsage.py:377-394).

    python examples/balance_switcher/train_switcher.py --model pinsage --dataset-path /tmp/ds/uk \\
        --num-sample-worker 3 --num-train-worker 5 --cache-percentage 0.2 --switch-cache-percentage 0.1
    python examples/balance_switcher/train_switcher.py --make-dataset small --single-gpu --no-switcher   # A/B
"""
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
sys.path.insert(0, os.path.join(ROOT, "examples", "multi_gpu"))
import samgraph.torch as sam  # noqa: E402
from models import MODELS  # noqa: E402
import train_fgnn  # noqa: E402  (arguments, run config, dataset writer of the multi-GPU example)


def parse_args():
    ap = train_fgnn.build_parser()
    ap.set_defaults(model="pinsage")
    ap.add_argument("--switch-cache-percentage", type=float, default=0.0,
                    help="feature cache of a switcher (its GPU also holds the graph; train_pinsage.py:115-116)")
    ap.add_argument("--no-switcher", action="store_true", help="the same run without switcher processes (A/B)")
    return ap.parse_args()


def run_sample(worker_id, rc):
    barrier, sem, stop = rc["global_barrier"], rc["mq_sem"], rc["sampler_stop_event"][worker_id]
    sam.sample_init(worker_id, rc["sample_workers"][worker_id])
    sam.notify_sampler_ready(barrier)
    num_epoch, num_step = sam.num_epoch(), sam.num_local_step()
    barrier.wait()  # run start
    times = []
    for epoch in range(num_epoch):
        tic = time.time()
        stop.clear()
        for _ in range(num_step):  # the epoch's permits: consumers block in the queue until the batches exist
            sem.release()
        barrier.wait()  # epoch start
        for _ in range(num_step):
            sam.sample_once()
        stop.set()  # this GPU is free: its switcher starts training
        times.append(time.time() - tic)
        barrier.wait()  # epoch end
    barrier.wait()  # results
    if worker_id == 0:
        print("test_result:sample_time={:.4f}".format(float(np.mean(times[1:])) if len(times) > 1 else times[0]))
    sam.shutdown()


def run_consume(worker_id, rc, is_switcher):
    barrier, sem, lock = rc["global_barrier"], rc["mq_sem"], rc["global_lock"]
    shared = rc["global_cpu_model"]
    name = "Switcher" if is_switcher else "Trainer"
    ctx = rc["sample_workers" if is_switcher else "train_workers"][worker_id]
    dev = torch.device(ctx)
    torch.cuda.set_device(dev)
    sam.wait_for_sampler_ready(barrier)  # (the pre-sampling ranking is the samplers')
    if is_switcher:
        sam.switch_init(worker_id, ctx, rc["switch_cache_percentage"])
    else:
        sam.train_init(worker_id, ctx)
    num_layer = rc["num_layer"]
    model = MODELS[rc["model"]](sam.feat_dim(), rc["num_hidden"], sam.num_class(), num_layer, rc["dropout"])
    model.load_state_dict(shared.state_dict())
    model = model.to(dev)
    params = list(model.parameters())
    shared_params = list(shared.parameters())
    loss_fcn = nn.CrossEntropyLoss().to(dev)
    opt = torch.optim.Adam(params, lr=rc["lr"])
    num_epoch = sam.num_epoch()
    get_blocks = sam.get_dgl_blocks_with_weights if rc["model"] == "pinsage" else sam.get_dgl_blocks
    model.train()
    barrier.wait()  # run start
    totals, counts, loss = [], [], None
    for epoch in range(num_epoch):
        barrier.wait()  # epoch start
        tic = time.time()
        if is_switcher:
            rc["sampler_stop_event"][worker_id].wait()
        n = 0
        while sem.acquire(timeout=0.01):
            sam.sample_once()
            key = sam.get_next_batch()
            blocks, batch_input, batch_label = get_blocks(key, num_layer)
            before = [p.detach().clone() for p in params]
            loss = loss_fcn(model(blocks, batch_input), batch_label)
            opt.zero_grad()
            loss.backward()
            opt.step()
            with lock, torch.no_grad():  # fold this step's delta into the job's model, take the result back
                for p, b, c in zip(params, before, shared_params):
                    p.copy_(c.to(dev) + (p - b))
                    c.copy_(p)
            torch.cuda.current_stream().synchronize()  # the batch's buffers go back to the pool at the next dequeue
            n += 1
        totals.append(time.time() - tic)
        counts.append(n)
        print("[{} {}] Epoch {:03d} | {:d} batches | {:.4f} s | loss {}".format(
            name, worker_id, epoch, n, totals[-1], "%.4f" % float(loss.detach()) if loss is not None else "-"), flush=True)
        barrier.wait()  # epoch end
    barrier.wait()  # results
    with rc["consumed"].get_lock():
        rc["consumed"].value += sum(counts)
        if is_switcher:
            rc["switched"].value += sum(counts)
    if not is_switcher and worker_id == 0:
        sl = slice(1, None) if len(totals) > 1 else slice(None)
        print("test_result:epoch_time:total={:.4f}".format(float(np.mean(totals[sl]))))
    sam.shutdown()


def main():
    args = parse_args()
    train_fgnn.make_dataset(args)
    rc = train_fgnn.get_run_config(args)
    ns, nt = rc["num_sample_worker"], rc["num_train_worker"]
    switchers = 0 if args.no_switcher else ns
    rc.update(have_switcher=1, switch_cache_percentage=args.switch_cache_percentage)
    sam.config({k: v for k, v in rc.items() if isinstance(v, (int, float, str, list)) and k not in
                train_fgnn.SCRIPT_KEYS + ("switch_cache_percentage",)})
    sam.data_init()  # before fork: nothing here touches the GPU
    # (one CPU thread: an OpenMP pool created here would not survive the fork -- a child's first parallel region on the
    # CPU, the copy of a large parameter into the shared model, would wait for threads that do not exist in it)
    torch.set_num_threads(1)
    torch.manual_seed(0)
    shared = MODELS[rc["model"]](sam.feat_dim(), rc["num_hidden"], sam.num_class(), rc["num_layer"], rc["dropout"])
    shared.share_memory()
    ctx = mp.get_context("fork")
    rc["global_cpu_model"] = shared
    rc["global_lock"] = ctx.Lock()
    rc["global_barrier"] = ctx.Barrier(ns + switchers + nt)
    rc["mq_sem"] = ctx.Semaphore(0)
    rc["sampler_stop_event"] = [ctx.Event() for _ in range(ns)]
    rc["consumed"], rc["switched"] = ctx.Value("l", 0), ctx.Value("l", 0)
    procs = [ctx.Process(target=run_sample, args=(i, rc)) for i in range(ns)]
    procs += [ctx.Process(target=run_consume, args=(i, rc, True)) for i in range(switchers)]
    procs += [ctx.Process(target=run_consume, args=(i, rc, False)) for i in range(nt)]
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
    total = sam.num_epoch() * sam.steps_per_epoch()
    print("test_result:batches_consumed={:d}".format(rc["consumed"].value))
    print("test_result:batches_by_switchers={:d}".format(rc["switched"].value))
    if rc["consumed"].value != total:
        sys.exit("consumed %d batches of %d" % (rc["consumed"].value, total))


if __name__ == "__main__":
    main()
