#!/usr/bin/env python3
"""the train_leg field of bench.py lines: show_train_leg.py <line.json> ..."""
import json
import sys

for f in sys.argv[1:]:
    d = json.loads(open(f).read().strip().splitlines()[-1])
    t = d["train_leg"]
    print(f, t.get("ms_per_step"), "| regions", t.get("timed_regions_run"), "late captures",
          t.get("graphs_captured_inside_the_reported_region"), "|", t.get("step"), "|", t.get("gemm_tuning"),
          t.get("error"), "| step", d["ms_per_step"])
