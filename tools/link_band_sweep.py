#!/usr/bin/env python3
"""Shape of the one-launch trainer-side extraction (fgnn_extract_fused) on a GPU that only extracts: workgroups of the
link band x loads in flight per lane, against the two-launch path it replaces (miss gather, then hit gather).

A papers100M-shaped batch per launch (302 K input rows of 512 B; --hit-rate of them from an 11 GB HBM cache, the rest
from a pinned host table of 2^24 rows), --streams batches in flight like an arch5 trainer's extraction thread.

  python3 tools/link_band_sweep.py [--hit-rate 0.905] [--batches 120] [--streams 4] [--out gpurun_out/x.json]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fgnn-artifacts_amd"))
from fgnn_hip import lib  # noqa: E402

lib.use_library(lib.PROF_LIB_PATH)  # FGNN_FUSED_LINK_UNROLL is read by the profiling build only


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--hit-rate", type=float, default=0.905)
    ap.add_argument("--rows", type=int, default=302000)
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--batches", type=int, default=120)
    ap.add_argument("--streams", type=int, default=4)
    ap.add_argument("--bands", default="1,2,4,8,16,32,64,128,256,512")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    lib.load()
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    n_host, n_cache = 1 << 24, 22_000_000
    host = torch.empty((n_host, a.dim), dtype=torch.float32).pin_memory()
    host.view(-1)[::4096] = 1.0  # touch every page
    cache = torch.empty((n_cache, a.dim), dtype=torch.float32, device=dev)
    n_miss = int(a.rows * (1 - a.hit_rate))
    n_hit = a.rows - n_miss
    NB = 2 * a.streams
    bufs = []
    for _ in range(NB):
        perm = torch.randperm(a.rows, generator=g, device=dev).to(torch.int32)
        bufs.append(dict(out=torch.empty((a.rows, a.dim), dtype=torch.float32, device=dev),
                         ms=torch.randint(0, n_host, (n_miss,), generator=g, device=dev, dtype=torch.int32),
                         md=perm[:n_miss].contiguous(),
                         cs=torch.randint(0, n_cache, (n_hit,), generator=g, device=dev, dtype=torch.int32),
                         cd=perm[n_miss:].contiguous()))
    streams = [torch.cuda.Stream(device=dev) for _ in range(a.streams)]
    row_b = a.dim * 4

    def run(launch, batches):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(batches):
            with torch.cuda.stream(streams[i % a.streams]):
                launch(bufs[i % NB])
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / batches * 1e3

    for b in bufs:  # per-workgroup start / end clocks of the launch (pinned: the kernel posts them, nothing is copied)
        b["stamps"] = torch.zeros(2 * 4096, dtype=torch.int64).pin_memory()
    shape = {}

    def fused(link):
        def f(b):
            shape["grid"], shape["link"] = lib.extract_fused(b["out"], host, cache, b["ms"], b["md"], b["cs"], b["cd"],
                                                             link_workgroups=link, stamps=b["stamps"])
        return f

    def band_ms():
        """(link band, HBM band) duration of the buffers' last launches: first start .. last end (100 MHz clock)"""
        ln, hb = [], []
        for b in bufs:
            st = b["stamps"][:2 * shape["grid"]].view(-1, 2).numpy()
            for lo, hi, acc in ((0, shape["link"], ln), (shape["link"], shape["grid"], hb)):
                if hi > lo:
                    acc.append((st[lo:hi, 1].max() - st[lo:hi, 0].min()) * 1e-5)
        return float(np.mean(ln)) if ln else 0.0, float(np.mean(hb)) if hb else 0.0

    def two_launch(shared):
        def f(b):
            lib.gather_rows(b["out"], host, src_index=b["ms"], dst_index=b["md"], src_row_mask=0xFFFFFFFF, shared_gpu=shared)
            lib.gather_rows(b["out"], cache, src_index=b["cs"], dst_index=b["cd"])
        return f

    res = []

    def report(name, ms, bands=None):
        gbs = n_miss * row_b / (ms * 1e-3) / 1e9
        res.append({"variant": name, "ms_per_batch": ms, "link_GBps": gbs, "link_frac_of_64": gbs / 64.0, "band_ms": bands})
        extra = ""
        if bands:
            hbm_gbs = n_hit * (2 * row_b + 8) / (bands[1] * 1e-3) / 1e9 if bands[1] else 0.0
            extra = "   link band %.3f ms, HBM band %.3f ms = %.0f GB/s" % (bands[0], bands[1], hbm_gbs)
        print("%-34s %.4f ms/batch   link %.1f GB/s (%.2f of 64)%s" % (name, ms, gbs, gbs / 64.0, extra), flush=True)

    print("# %d rows/batch, hit rate %.3f: %d miss rows (%.1f MB over the link), %d hit rows; %d batches in flight"
          % (a.rows, a.hit_rate, n_miss, n_miss * row_b / 1e6, n_hit, a.streams), flush=True)
    for shared in (1, 0):
        f = two_launch(shared)
        run(f, 16)
        report("two launches (host grid %s)" % ("64" if shared else "HBM-sized"), min(run(f, a.batches) for _ in range(3)))
    for ul in (8, 4):
        os.environ["FGNN_FUSED_LINK_UNROLL"] = str(ul)
        for band in [int(x) for x in a.bands.split(",")]:
            f = fused(band)
            run(f, 16)
            t = min(run(f, a.batches) for _ in range(3))
            report("fused band=%d unroll=%d" % (band, ul), t, band_ms())
    # the band alone (no hit rows): what the link gives this access pattern
    os.environ["FGNN_FUSED_LINK_UNROLL"] = "8"
    for band in (64, 128, 256):
        f = lambda b, band=band: lib.extract_fused(b["out"], host, cache, b["ms"], b["md"], None, None, link_workgroups=band)
        run(f, 16)
        report("miss rows only band=%d" % band, min(run(f, a.batches) for _ in range(3)))
    if a.out:
        with open(a.out, "w") as fh:
            json.dump({"args": vars(a), "results": res}, fh, indent=1)


if __name__ == "__main__":
    main()
