#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace CSV of the overlapped bench: what bounds the steady-state period of a batch.
khop2's in-place CSR swaps keep the sampler kernels of consecutive batches in order (engine.hip), so the chain
  sampler(layer 1) -> dedup count+assign -> [remap fix-up] -> sampler(layer 0)  -> next batch's sampler(layer 1)
is a lower bound of the period.  Prints, over the steady middle of the run: the period (start to start of consecutive
layer-1 samplers), the average duration of every kernel kind, and the idle gaps on that chain.
usage: chain_period.py <dir or kernel_trace.csv>"""
import csv
import glob
import re
import sys
from collections import defaultdict

p = sys.argv[1]
f = p if p.endswith(".csv") else glob.glob(p + "/**/*kernel_trace.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    m = re.search(r"fgnn::(?:\(anonymous namespace\)::)?(\w+)", n) or re.search(r"(__amd_rocclr_copyBuffer)", n)
    if not m:
        continue
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(1), r.get("Stream_Id", r.get("Queue_Id"))))
rows.sort()
samp = [r for r in rows if r[2].startswith("khop_sample")]
# layers alternate 1, 0, 1, 0 ... in start order (ordered chain); drop the first/last quarter
n = len(samp) // 2
lo, hi = n // 4, n - n // 4
l1 = [samp[2 * i] for i in range(lo, hi)]
l0 = [samp[2 * i + 1] for i in range(lo, hi)]
period = (l1[-1][0] - l1[0][0]) / (len(l1) - 1) / 1e3
print("batches analysed %d   period (L1 sampler start to start) %.1f us" % (len(l1), period))
print("  L1 sampler %.1f us | L1 end -> L0 start %.1f us | L0 sampler %.1f us | L0 end -> next L1 start %.1f us" % (
    sum(e - s for s, e, _, _ in l1) / len(l1) / 1e3,
    sum(b[0] - a[1] for a, b in zip(l1, l0)) / len(l1) / 1e3,
    sum(e - s for s, e, _, _ in l0) / len(l0) / 1e3,
    sum(b[0] - a[1] for a, b in zip(l0, l1[1:])) / (len(l1) - 1) / 1e3))
t0, t1 = l1[0][0], l1[-1][0]
dur = defaultdict(list)
for s, e, name, q in rows:
    if t0 <= s < t1:
        dur[name].append((e - s) / 1e3)
print("kernel kind: launches per batch, avg us, total us per batch")
tot = 0.0
for name, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    per = sum(v) / (len(l1) - 1)
    tot += per
    print("  %-28s %5.2f  %7.1f  %7.1f" % (name, len(v) / (len(l1) - 1), sum(v) / len(v), per))
print("sum of kernel time per batch %.1f us = %.2f x period" % (tot, tot / period))
