#!/usr/bin/env python3
"""Consumption rate along a pipeline span (bench.py --gpus N with FGNN_BENCH_DUMP_STAMPS=file): mean ms per batch over
consecutive groups of G consumed batches.  usage: stamp_profile.py file [G=20]"""
import sys
rows = [ln.split() for ln in open(sys.argv[1]) if not ln.startswith("#")]
t = [float(r[0]) for r in rows]
G = int(sys.argv[2]) if len(sys.argv) > 2 else 20
print("# batches  t_begin_s  ms_per_batch")
for a in range(0, len(t) - G, G):
    print("%5d-%-5d %8.4f  %.4f" % (a, a + G - 1, t[a], (t[a + G] - t[a]) / G * 1e3))
