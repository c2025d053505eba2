#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace CSV of the default (two-stream) bench: how busy is the GPU in the timed region,
how much do the streams overlap.  usage: overlap_stats.py <dir>"""
import csv
import glob
import re
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    m = re.search(r"(?:fgnn|sam)::(?:\(anonymous namespace\)::)?(\w+)", n) or re.search(r"(__amd_rocclr_copyBuffer)", n)
    if not m:
        continue
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(1), r.get("Stream_Id", r.get("Queue_Id"))))
rows.sort()
first = next(i for i, r in enumerate(rows) if r[2].startswith("khop_sample"))
starts = [i for i, r in enumerate(rows) if i > first and r[2] == "__amd_rocclr_copyBuffer"]  # one per batch (its end)
if len(starts) < 151:  # the summary copy rides on the batch's last kernel (GatherTail / pack kernel)
    starts = [i for i, r in enumerate(rows) if i > first and r[2] in ("gather_rows16_kernel", "pack_kernel")]
a, b = starts[30], starts[150]  # inside the timed region
seg = rows[a:b]
t0, t1 = seg[0][0], max(r[1] for r in seg)
ev = []
for s, e, _, _ in seg:
    ev.append((s, 1))
    ev.append((e, -1))
ev.sort()
busy = over = 0
depth = 0
last = t0
for t, d in ev:
    if depth >= 1:
        busy += t - last
    if depth >= 2:
        over += t - last
    depth += d
    last = t
span = t1 - t0
print("batches %d  span %.1f us  per batch %.1f us" % (120, span / 1e3, span / 1e3 / 120))
print("GPU busy (>=1 kernel) %.1f%%   >=2 kernels running %.1f%%   idle %.1f%%" % (
    100.0 * busy / span, 100.0 * over / span, 100.0 * (span - busy) / span))
per = {}
for s, e, n, q in seg:
    per.setdefault(q, 0)
    per[q] += e - s
print("kernel time per stream / span:", {q: round(v / span, 3) for q, v in per.items()})
