#!/usr/bin/env python3
"""Kernel timeline of a window of a rocprofv3 --kernel-trace CSV with overlapping streams: every kernel of the window
with start, duration, stream.  usage: overlap_timeline.py <dir> <skip_us> <window_us>"""
import csv
import glob
import re
import sys

d, skip, win = sys.argv[1], float(sys.argv[2]), float(sys.argv[3])
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    m = re.search(r"(?:fgnn|sam)::(?:\(anonymous namespace\)::)?(\w+)", n) or re.search(r"(__amd_rocclr_\w+)", n)
    if m:
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(1), r.get("Stream_Id", r.get("Queue_Id", "?"))))
rows.sort()
last = rows[-1][1]
t0 = last - int((skip + win) * 1e3)
print("start_us  dur_us  stream  kernel   (window of %.0f us ending %.0f us before the last kernel)" % (win, skip))
for s, e, n, q in rows:
    if s >= t0 and s < t0 + win * 1e3:
        print("%8.1f %7.1f  %s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, n))
