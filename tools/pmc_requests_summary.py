#!/usr/bin/env python3
"""tools/pmc_requests.sh's pass directories -> JSON: per workload and kernel, fabric read / write requests per batch
(TCC_EA0_RDREQ_sum / TCC_EA0_WRREQ_sum: what the XCDs' L2s send to the memory side, 32- or 64-byte requests), and the
requests ONE read of bench.py's random-read probe makes in the same pass.  usage: pmc_requests_summary.py <dir> <workloads...>"""
import csv
import glob
import json
import re
import sys
from collections import defaultdict

O = sys.argv[1]
GATHER = ("gather_rows16_kernel", "gather_rows_elem_kernel")
out = {"command": "rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --kernel-trace --output-format csv -- python3 bench.py "
                  "--workload <w> --steps 20 --warmup 3 --windows 1 --no-overlap --no-cpu-baseline --timed-only",
       "unit": "fabric requests (L2 -> memory side) per batch; sampler side = every kernel of a batch but the feature gather",
       "workloads": {}}
for wl in sys.argv[2:]:
    fs = glob.glob("%s/%s_req/**/*counter_collection.csv" % (O, wl), recursive=True)
    if not fs:
        continue
    acc = defaultdict(lambda: defaultdict(float))
    launches = defaultdict(set)
    for r in csv.DictReader(open(fs[0])):
        n = r["Kernel_Name"]
        m = re.search(r"(?:fgnn|sam)::(?:\(anonymous namespace\)::)?(\w+)", n) or re.search(r"(random_read_probe_kernel)", n)
        if not m:
            continue
        k = m.group(1)
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        launches[k].add(r.get("Dispatch_Id") or r.get("Correlation_Id") or len(launches[k]))
    gk = next((k for k in acc if k in GATHER), None)
    batches = len(launches[gk]) if gk else 1
    # the warm-up's probe launches (lib.random_read_rate: 1 + 24 launches of 4 M reads each)
    probe = acc.get("random_read_probe_kernel")
    probe_reads = len(launches.get("random_read_probe_kernel", ())) * 4_000_000
    kernels = {}
    side = {"read": 0.0, "write": 0.0}
    for k in acc:
        if k == "random_read_probe_kernel":
            continue
        rd, wr = acc[k].get("TCC_EA0_RDREQ_sum", 0.0) / batches, acc[k].get("TCC_EA0_WRREQ_sum", 0.0) / batches
        per_batch = len(launches[k]) / batches
        if per_batch < 0.5:  # set-up kernels (search trees, table wipes): not part of a batch
            continue
        kernels[k] = {"launches_per_batch": per_batch, "read_per_batch": rd, "write_per_batch": wr}
        if k not in GATHER:
            side["read"] += rd
            side["write"] += wr
    out["workloads"][wl] = {"batches": batches, "kernels": kernels, "sampler_side_per_batch": side,
                            "probe_requests_per_read": (probe.get("TCC_EA0_RDREQ_sum", 0.0) / probe_reads) if probe and probe_reads else None}
print(json.dumps(out, indent=1))
