#!/usr/bin/env python3
"""Rate of the miss-row path: random feature rows gathered by the GPU straight from pinned HOST memory
(fgnn_gather_rows with a host source -- what the trainer-side extraction does for cache misses, instead of the
reference's OpenMP gather + H2D copy, cuda_cache_manager_host.cc:38-56) against a plain pinned H2D copy of the same
number of bytes.  usage: host_gather.py [rows_in_table] [rows_per_gather]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fgnn-artifacts_amd"))
from fgnn_hip import lib  # noqa: E402


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def main():
    n_table = int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000
    n_rows = int(sys.argv[2]) if len(sys.argv) > 2 else 800_000
    lib.load()
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    for dim in (100, 128, 256):
        table = torch.empty((n_table, dim), dtype=torch.float32).pin_memory()
        table[:, 0] = torch.arange(n_table, dtype=torch.float32)
        idx = torch.randint(0, n_table, (n_rows,), generator=g, device=dev, dtype=torch.int32)
        out = torch.empty((n_rows, dim), dtype=torch.float32, device=dev)
        nbytes = n_rows * dim * 4
        for wg in ("4", "8", "16"):
            os.environ["FGNN_GATHER_WG_PER_CU"] = wg
            dt = timed(lambda: lib.gather_rows(out, table, src_index=idx))
            ok = bool((out[:, 0].long() == (idx.long() & 0xFFFFFFFF)).all())
            print("dim %3d  %d rows from a %.1f GB pinned table, %2s workgroups/CU: %.2f ms = %5.1f GB/s over the host "
                  "link  %s" % (dim, n_rows, n_table * dim * 4 / 1e9, wg, dt * 1e3, nbytes / dt / 1e9,
                                "ok" if ok else "WRONG"))
        os.environ.pop("FGNN_GATHER_WG_PER_CU")
        flat = table[:n_rows]
        dt = timed(lambda: out.copy_(flat, non_blocking=True))
        print("dim %3d  plain pinned H2D copy of the same %.0f MB: %.2f ms = %5.1f GB/s" % (dim, nbytes / 1e6, dt * 1e3,
                                                                                         nbytes / dt / 1e9))
        del table, out


if __name__ == "__main__":
    main()
