#!/usr/bin/env python3
"""Independent random 4-byte reads per second (fgnn_debug_random_reads) against the footprint they fall into: what a
look-up structure gains by fitting the chip's caches (4 MB of L2 per XCD, 256 MB of memory-side cache) instead of
spanning HBM.

  python3 tools/probe_footprint.py [--out gpurun_out/x.txt]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fgnn-artifacts_amd"))
from fgnn_hip import lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    big = torch.zeros(1 << 30, dtype=torch.int32, device=dev)  # 4 GiB
    lines = ["# footprint  random 4-byte reads / s  (4 M reads per launch, 24 launches; 3 runs)"]
    for mb in (1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 4096):
        n = mb << 18
        r = [lib.random_read_rate(big[:n]) for _ in range(3)]
        lines.append("%5d MB   %s G/s" % (mb, "  ".join("%6.1f" % (x / 1e9) for x in r)))
        print(lines[-1], flush=True)
    if a.out:
        with open(a.out, "w") as f:
            f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
