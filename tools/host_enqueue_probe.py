#!/usr/bin/env python3
"""Where does the host time of a batch go, and does it depend on which CPUs the enqueueing thread runs on?

Prints the CPUs this process may use, the GPU's NUMA node and the node of every allowed CPU, then -- for the allowed
CPUs as given, for those on the GPU's node and for those on other nodes -- the host time per batch of the native batch
loop (fgnn_sampler_run_range, papers100M-like batches on a smaller R-MAT graph) and the cost of a bare kernel launch.
  python3 tools/host_enqueue_probe.py [--steps 200]"""
import argparse
import ctypes as C
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import bench  # noqa: E402
from fgnn_hip import lib, rmat  # noqa: E402


def node_of_cpu():
    out = {}
    base = "/sys/devices/system/node"
    try:
        for d in os.listdir(base):
            if d.startswith("node") and d[4:].isdigit():
                for part in open(f"{base}/{d}/cpulist").read().strip().split(","):
                    if part:
                        a, _, b = part.partition("-")
                        for c in range(int(a), int(b or a) + 1):
                            out[c] = int(d[4:])
    except Exception as e:
        print("no NUMA info:", e)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    L = lib.load()
    allowed = sorted(os.sched_getaffinity(0))
    n2c = node_of_cpu()
    gnode = bench.gpu_numa_node(0)
    by_node = {}
    for c in allowed:
        by_node.setdefault(n2c.get(c, -1), []).append(c)
    print("allowed CPUs:", len(allowed), "by node:", {k: (len(v), v[:4]) for k, v in by_node.items()}, "GPU node:", gnode,
          "nodes on host:", sorted(set(n2c.values())))
    num_node, num_edge, bs, fan = 8_000_000, 128_000_000, 8000, [25, 10]
    indptr, indices, _ = rmat.rmat_csr(num_node, num_edge, 42, dev)
    train = rmat.train_set(num_node, 1_000_000, 1, dev)
    feat = torch.zeros((num_node, 128), dtype=torch.float32, device=dev)
    label = torch.zeros((num_node,), dtype=torch.int64, device=dev)
    table = torch.full((num_node,), -1, dtype=torch.int32, device=dev)
    table[::5] = torch.arange((num_node + 4) // 5, device=dev, dtype=torch.int32)
    sampler = lib.Sampler(indptr, indices, fan, bs, sample_type=lib.KHOP2)
    streams = [torch.cuda.Stream(device=dev) for _ in range(3)]
    batches = [sampler.new_batch(128, lib.F32, lib.I64) for _ in range(6)]
    torch.cuda.synchronize()
    seq = [0]

    def measure(tag):
        sampler.run_range(seq[0], 30, train, bs, batches, streams, cache_table=table, feat=feat, label=label)
        seq[0] += 30
        res = []
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            _, _, busy = sampler.run_range(seq[0], a.steps, train, bs, batches, streams, cache_table=table, feat=feat,
                                           label=label)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            seq[0] += a.steps
            res.append((dt / a.steps * 1e3, busy / a.steps * 1e3))
        # bare launches: one workgroup that exits at once, 2000 launches on one stream
        st = C.c_void_p(streams[0].cuda_stream)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(2000):
            L.fgnn_debug_occupy(C.c_size_t(1), C.c_uint(0), st)
        t_launch = (time.perf_counter() - t0) / 2000 * 1e6
        torch.cuda.synchronize()
        print("%-28s cpu now %3d | ms/step %s | host enqueue ms/step %s | bare launch via ctypes %.2f us" % (
            tag, C.CDLL(None).sched_getcpu(), " ".join("%.4f" % r[0] for r in res), " ".join("%.4f" % r[1] for r in res), t_launch))

    measure("as given")
    for node, cpus in sorted(by_node.items()):
        os.sched_setaffinity(0, cpus)
        time.sleep(0.05)
        measure("node %d%s (%d cpus)" % (node, " = GPU's" if node == gnode else "", len(cpus)))
        os.sched_setaffinity(0, cpus[:1])
        time.sleep(0.05)
        measure("node %d, one cpu (%d)" % (node, cpus[0]))
    os.sched_setaffinity(0, allowed)


if __name__ == "__main__":
    main()
