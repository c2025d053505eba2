#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs as the MI355X guide
prescribes) of `bench.py --no-overlap`.  usage: pmc_summary.py <dir FETCH_SIZE> <dir WRITE_SIZE> [bench log with the JSON line]
gfx950 correction: FETCH_SIZE counts 128-B read requests at 64 B for wide coalesced reads -> doubled for the gather."""
import csv
import glob
import json
import re
import sys
from collections import defaultdict


def load(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    acc, cnt = defaultdict(float), defaultdict(int)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        n = r["Kernel_Name"]
        m = re.search(r"fgnn::(?:\(anonymous namespace\)::)?(\w+(?:<[^>]*>)?)", n)
        if not m:
            continue
        acc[m.group(1)] += float(r["Counter_Value"])
        cnt[m.group(1)] += 1
    return {k: (acc[k] / cnt[k], cnt[k]) for k in acc}


fetch = load(sys.argv[1], "FETCH_SIZE")
write = load(sys.argv[2], "WRITE_SIZE")
out = {"command": "rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -- python3 bench.py --steps 20 --warmup 3 "
                  "--no-cpu-baseline --no-overlap (second pass: --pmc WRITE_SIZE)",
       "unit": "KB per launch (rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KB)"}
gk = [k for k in fetch if k.startswith("gather_rows16_kernel")][0]
out["kernel"] = gk
out["FETCH_SIZE_KB_per_launch"] = fetch[gk][0]
out["WRITE_SIZE_KB_per_launch"] = write[gk][0]
out["correction"] = ("gfx950: FETCH_SIZE counts 128-B read requests at 64 B for wide (16 B/lane) coalesced reads -> doubled "
                     "(MI355X_MICROARCH.md, HBM); WRITE_SIZE exact for 16 B/lane stores")
hbm = (2 * fetch[gk][0] + write[gk][0]) * 1024.0
out["hbm_bytes_per_launch"] = hbm
if len(sys.argv) > 3:
    line = [l for l in open(sys.argv[3]) if l.startswith("{")][-1]
    j = json.loads(line)
    alg = j["roofline"]["algorithmic_bytes_per_launch"]
    out["algorithmic_bytes_per_launch"] = alg
    out["traffic_over_algorithmic"] = hbm / alg
out["all_kernels_KB_per_launch"] = {k: {"launches": fetch[k][1], "FETCH_SIZE": fetch[k][0],
                                        "WRITE_SIZE": write.get(k, (0, 0))[0]} for k in sorted(fetch)}
# every kernel of the chain against the algorithmic bytes of the stage it implements (SURVEY.md 8(d) figures as
# bench.algorithmic_bytes computes them, per batch): the sampler launches carry the dedup's insert, so sampling and
# dedup/remap are one stage here.  FETCH_SIZE is doubled only for the wide coalesced reads of the row gather; the
# random 4/8-byte reads of the other kernels are 64-B requests and count as reported.
STAGES = {"sample_dedup_remap": ("khop_sample_kernel", "ht_count_assign_kernel", "ht_map_fix_kernel", "ht_insert_kernel", "part_hist_kernel", "part_scatter_kernel", "part_dedup_kernel",
                                 "weighted_", "rank_", "hash_dedup_kernel", "random_walk_topk_kernel", "rw_emit_kernel"),
          "cache_split": ("cache_split_fused_kernel", "cache_count_kernel", "cache_split_kernel"),
          "gather": ("gather_rows16_kernel", "gather_rows_elem_kernel")}
if len(sys.argv) > 3:
    batches = fetch[gk][1]
    ab = j.get("algorithmic_bytes_per_step", {})
    alg = {"sample_dedup_remap": ab.get("sample", 0) + ab.get("dedup_remap", 0), "cache_split": ab.get("cache_split", 0),
           "gather": ab.get("gather", 0)}
    per_kernel, per_stage = {}, {}
    for k in sorted(fetch):
        wide = 2.0 if k.startswith("gather_rows16_kernel") else 1.0
        b = (wide * fetch[k][0] + write.get(k, (0, 0))[0]) * 1024.0 * fetch[k][1] / batches
        stage = next((s for s, pre in STAGES.items() if k.startswith(pre)), None)
        per_kernel[k] = {"stage": stage, "launches_per_batch": fetch[k][1] / batches, "hbm_bytes_per_batch": b}
        if stage:
            per_stage.setdefault(stage, 0.0)
            per_stage[stage] += b
    out["per_kernel"] = per_kernel
    out["per_stage"] = {s: {"hbm_bytes_per_batch": v, "algorithmic_bytes_per_batch": alg[s],
                            "traffic_over_algorithmic": v / alg[s] if alg[s] else None} for s, v in per_stage.items()}
print(json.dumps(out, indent=1))
