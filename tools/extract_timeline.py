#!/usr/bin/env python3
"""Kernel timeline of the N = 1 extract leg (features in host memory) from a rocprofv3 --kernel-trace CSV of bench.py:
the window spanned by the last few HOST-source gathers (gather launches longer than 150 us), every kernel with start,
duration and stream, plus how many host gathers run at once.  usage: extract_timeline.py <dir> [num_batches]"""
import csv
import glob
import re
import sys

d = sys.argv[1]
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 6
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    m = re.search(r"(?:fgnn|sam)::(?:\(anonymous namespace\)::)?(\w+)", n) or re.search(r"(__amd_rocclr_\w+)", n)
    if m:
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(1), r.get("Stream_Id", r.get("Queue_Id", "?"))))
rows.sort()
host = [(s, e) for s, e, n, q in rows if n == "gather_rows16_kernel" and e - s > 150000]
if len(host) < nb + 2:
    sys.exit("no extract leg in the trace (%d long gathers)" % len(host))
t0, t1 = host[-nb - 1][0], host[-1][1]
print("extract leg, last %d batches: window %.1f us = %.1f us per batch" % (nb, (t1 - t0) / 1e3, (t1 - t0) / 1e3 / (nb + 1)))
print("start_us  dur_us  stream  kernel")
for s, e, n, q in rows:
    if s >= t0 and s <= t1:
        print("%8.1f %7.1f  %s  %s%s" % ((s - t0) / 1e3, (e - s) / 1e3, q, n, "   <-- host rows" if (n == "gather_rows16_kernel" and e - s > 150000) else ""))
# link occupancy: time with >= 1 / >= 2 host gathers running
ev = sorted([(s, 1) for s, e in host if s >= t0] + [(e, -1) for s, e in host if s >= t0])
cur, last, busy1, busy2 = 0, t0, 0, 0
for t, dlt in ev:
    if cur >= 1:
        busy1 += t - last
    if cur >= 2:
        busy2 += t - last
    cur += dlt
    last = t
print("host gathers running: >= 1 for %.0f %% of the window, >= 2 for %.0f %%" % (100 * busy1 / (t1 - t0), 100 * busy2 / (t1 - t0)))
