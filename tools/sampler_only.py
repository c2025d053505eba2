#!/usr/bin/env python3
"""Throughput of ONE arch5 sampler process with nobody consuming (an epoch fits in the 170-slot queue): sample + cache
index + serialisation into the hand-off ring, i.e. what a sampler GPU can feed its trainers.
usage: SAMGRAPH_EMPTY_FEAT=24 [SAMGRAPH_DEVICE_RING_SLOTS=170] sampler_only.py <dataset dir> [cache_percentage]"""
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fgnn-artifacts_amd"))
import samgraph.torch as sam  # noqa: E402


def sampler(q):
    sam.sample_init(0, "cuda:0")
    n = sam.num_local_step()
    t0 = time.time()
    for _ in range(n):
        sam.sample_once()
    dt = time.time() - t0
    edges = sum(sam.get_log_step_value(0, s, sam.kLogL1NumSample) for s in range(n))
    q[0], q[1], q[2] = n, dt, edges
    os._exit(0)  # nobody reads the queue: skip the drain of shutdown


if __name__ == "__main__":
    path, cache = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 0.2
    sam.config(dict(dataset_path=path, _arch=sam.kArch5, _sample_type=sam.sample_types["khop2"], batch_size=8000,
                    num_epoch=1, _cache_policy=sam.cache_policies["pre_sample"], presample_epoch=1, cache_percentage=cache,
                    max_sampling_jobs=10, max_copying_jobs=2, omp_thread_num=8, num_sample_worker=1,
                    num_train_worker=1, num_fanout=2, fanout=[25, 10]))
    sam.data_init()
    ctx = mp.get_context("fork")
    q = ctx.Array("d", 3)
    p = ctx.Process(target=sampler, args=(q,))
    p.start()
    p.join(timeout=200)
    if p.is_alive():
        p.terminate()
        sys.exit("sampler did not finish")
    n, dt, edges = int(q[0]), q[1], q[2]
    print("ring slots %s: %d batches in %.4f s = %.1f us per batch, %.3e sampled edges/s" % (
        os.environ.get("SAMGRAPH_DEVICE_RING_SLOTS", "0"), n, dt, dt / n * 1e6, edges / dt))
