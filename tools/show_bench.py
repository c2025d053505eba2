#!/usr/bin/env python3
"""prints the interesting fields of bench.py JSON lines: show_bench.py file..."""
import json
import sys


def short(x):
    if isinstance(x, dict):
        return {k: short(v) for k, v in x.items() if k not in ("all_runs", "config", "metric", "note", "sample", "traffic_source")}
    if isinstance(x, str):
        return x[:70]
    if isinstance(x, float):
        return float("%.4g" % x)
    return x


for f in sys.argv[1:]:
    try:
        d = json.loads([ln for ln in open(f).read().splitlines() if ln.startswith("{")][-1])
    except Exception as e:
        print(f, "unreadable:", e)
        continue
    print("==", f)
    print(json.dumps(short(d), indent=1))
