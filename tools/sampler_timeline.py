#!/usr/bin/env python3
"""One arch5 sampler process of the bench pipeline, in THIS process and with nobody consuming (an epoch fits in the
queue), so that `rocprofv3 --kernel-trace -- python3 tools/sampler_timeline.py` sees its kernels: where a sampler GPU's
time per batch goes (sampling chain, cache split, pack into the HBM ring) against the host's enqueue time.
Prints the per-batch wall time; SAMGRAPH_LOG_LEVEL=info adds the engine's own host-side split."""
import os
import shutil
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "fgnn-artifacts_amd"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # as samgraph_config would, had this process not touched the GPU before it
os.environ.setdefault("SAMGRAPH_DEVICE_RING_SLOTS", "170")
os.environ.setdefault("SAMGRAPH_DEVICE_RING_DRAIN_S", "0.01")
os.environ.setdefault("SAMGRAPH_EMPTY_FEAT", "24")
import torch  # noqa: E402

import bench  # noqa: E402

args = bench.parse_args(sys.argv[1:])
w = bench.WORKLOADS[args.workload]
args.sample_type = args.sample_type or w["sample_type"]
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
tag = "fgnn_st_%d" % os.getpid()
os.environ["SAMGRAPH_SHM_PREFIX"] = tag
out_dir = os.path.join("/dev/shm" if os.path.isdir("/dev/shm") else "/tmp", tag + "_ds")
try:
    bench.write_dataset(args, w, dev, out_dir)
    be = bench.EngineBackend(args, w, {"dir": out_dir}, 1, 1, True, 0, 0, 1)
    be.role_init()
    n = be.num_local_step()
    t0 = time.perf_counter()
    for _ in range(n):
        be.sample_once()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    edges = sum(be.sam.get_log_step_value(0, s, be.sam.kLogL1NumSample) for s in range(n))
    print("sampler alone: %d batches in %.4f s = %.1f us per batch, %.3e sampled edges/s" % (n, dt, dt / n * 1e6, edges / dt))
    sam = be.sam
    ts = [sam.get_log_step_value(0, s, sam.kLogL1SampleTime) for s in range(n)]
    ti = [sam.get_log_step_value(0, s, sam.kLogL3CacheGetIndexTime) for s in range(n)]
    print("per batch, from the kernels' own time stamps: sample %.1f us (median), cache index %.1f us" % (
        sorted(ts)[n // 2] * 1e6, sorted(ti)[n // 2] * 1e6))
    be.shutdown()
finally:
    shutil.rmtree(out_dir, ignore_errors=True)
    for f in os.listdir("/dev/shm") if os.path.isdir("/dev/shm") else []:
        if f.startswith(tag):
            try:
                os.unlink(os.path.join("/dev/shm", f))
            except OSError:
                pass
