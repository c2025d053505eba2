#!/usr/bin/env python3
"""Prints the kernel timeline of one steady-state batch from a rocprofv3 --kernel-trace CSV:
start offset, duration and the idle gap before each kernel.  usage: chain_timeline.py <dir> [batch_index]"""
import csv
import glob
import re
import sys

d = sys.argv[1]
which = int(sys.argv[2]) if len(sys.argv) > 2 else -10
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    m = re.search(r"fgnn::(?:\(anonymous namespace\)::)?(\w+)", n) or re.search(r"sam::(\w+)", n) or re.search(r"(__amd_rocclr_\w+)", n)
    if not m:
        continue
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(1), r.get("Stream_Id", r.get("Queue_Id", "?"))))
rows.sort()
# a batch ends with its feature gather (bench; the label rows and the summary copy ride on that launch) or with the pack
# kernel (an arch5 sampler); traces of older builds end a batch with the D2H copy of its summary
first = next(i for i, r in enumerate(rows) if r[2].startswith("khop_sample") or r[2].startswith("ht_start_batch"))
names = {r[2] for r in rows}
last = "gather_rows16_kernel" if "gather_rows16_kernel" in names else "pack_kernel" if "pack_kernel" in names \
    else "__amd_rocclr_copyBuffer"
ends = [i for i, r in enumerate(rows) if i > first and r[2] == last]
b = ends[which - 1] + 1
e = ends[which] + 1
t0 = rows[b][0]
prev_end = t0
busy = 0
print("batch timeline (us): start  dur  gap  stream  kernel")
for s, en, n, q in rows[b:e]:
    print("%8.1f %7.1f %6.1f  %s  %s" % ((s - t0) / 1e3, (en - s) / 1e3, (s - prev_end) / 1e3, q, n))
    busy += en - s
    prev_end = max(prev_end, en)
print("total span %.1f us, kernel busy %.1f us, kernels %d" % ((rows[e - 1][1] - t0) / 1e3, busy / 1e3, e - b))
