#!/usr/bin/env python3
"""Condenses a rocprofv3 --kernel-trace --stats output directory into a markdown table of this repo's kernels.
usage: stats_summary.py <dir> [steps]   (steps: divide total time by this to get us per batch)"""
import csv
import glob
import re
import sys

d = sys.argv[1]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 0
f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    n = r["Name"]
    m = re.search(r"fgnn::(?:\(anonymous namespace\)::)?(\w+(?:<[^>]*>)?)", n) or re.search(r"sam::(\w+)", n) or re.search(r"(__amd_rocclr_\w+)", n)
    if not m:
        continue
    rows.append((m.group(1), int(r["Calls"]), float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3,
                 float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
rows.sort(key=lambda x: -x[2])
print("| kernel | calls | avg us | min us | max us |" + (" us per batch |" if steps else ""))
print("|---|---|---|---|---|" + ("---|" if steps else ""))
for n, c, tot, avg, mn, mx in rows:
    print("| %s | %d | %.2f | %.1f | %.1f |" % (n, c, avg, mn, mx) + (" %.1f |" % (tot / steps) if steps else ""))
