#!/usr/bin/env python3
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
q = {}
for r in rows:
    q.setdefault(r["Queue_Id"], set()).add(r.get("Stream_Id", "?"))
print(sys.argv[1], "kernels", len(rows), "distinct queues", len(q), {k: sorted(v) for k, v in q.items()})
