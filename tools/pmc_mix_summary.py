#!/usr/bin/env python3
"""Condenses tools/pmc_mix.sh's directories: per kernel, mix (three streams) against serial (one stream) -- average
duration from the plain trace, counters per launch from the four --pmc passes, and whether kernels of different streams
still overlapped in each run (a counter pass that serialises the launches cannot show the mix).
usage: pmc_mix_summary.py <dir> <workload>"""
import csv
import glob
import re
import sys
from collections import defaultdict

O, wl = sys.argv[1], sys.argv[2]


def short(n):
    m = re.search(r"(?:fgnn|sam)::(?:\(anonymous namespace\)::)?(\w+)", n)
    return m.group(1) if m else None


def trace(d):
    fs = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
    if not fs:
        return {}, None
    rows = []
    for r in csv.DictReader(open(fs[0])):
        k = short(r["Kernel_Name"])
        if k:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), k))
    rows.sort()
    rows = rows[len(rows) // 3:]  # the steady part
    dur = defaultdict(list)
    for s, e, k in rows:
        dur[k].append((e - s) / 1e3)
    ev = sorted([(s, 1) for s, _, _ in rows] + [(e, -1) for _, e, _ in rows])
    busy = over = depth = 0
    last = ev[0][0]
    for t, d_ in ev:
        if depth >= 1:
            busy += t - last
        if depth >= 2:
            over += t - last
        depth += d_
        last = t
    return {k: (sum(v) / len(v), len(v)) for k, v in dur.items()}, (over / busy if busy else 0.0)


def counters(d):
    fs = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    acc, cnt = defaultdict(lambda: defaultdict(float)), defaultdict(lambda: defaultdict(int))
    if not fs:
        return {}
    for r in csv.DictReader(open(fs[0])):
        k = short(r["Kernel_Name"])
        if not k:
            continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k][r["Counter_Name"]] += 1
    return {k: {c: acc[k][c] / cnt[k][c] for c in acc[k]} for k in acc}


res = {}
for mode in ("mix", "serial"):
    dur, ov = trace("%s/%s_%s_trace" % (O, wl, mode))
    res[mode] = {"dur": dur, "overlap": ov, "pmc": {}, "pmc_overlap": {}}
    for st in ("SQ", "TCP", "TCC", "EA"):
        d = "%s/%s_%s_%s" % (O, wl, mode, st)
        for k, v in counters(d).items():
            res[mode]["pmc"].setdefault(k, {}).update(v)
        _, ov2 = trace(d)
        res[mode]["pmc_overlap"][st] = ov2
print("# %s: mix = bench.py's three batch streams, serial = one stream; plain kernel traces and four --pmc passes each" % wl)
print("# share of the GPU-busy time with >= 2 kernels running: plain trace mix %.2f, serial %.2f; inside the counter passes: mix %s, serial %s"
      % (res["mix"]["overlap"] or 0, res["serial"]["overlap"] or 0,
         {k: round(v, 2) if v is not None else None for k, v in res["mix"]["pmc_overlap"].items()},
         {k: round(v, 2) if v is not None else None for k, v in res["serial"]["pmc_overlap"].items()}))
kern = sorted(res["serial"]["dur"], key=lambda k: -res["serial"]["dur"][k][0] * res["serial"]["dur"][k][1])
print("\n## average kernel duration (us), plain traces")
print("%-32s %8s %10s %10s %7s" % ("kernel", "launches", "serial", "mix", "ratio"))
for k in kern:
    s = res["serial"]["dur"][k]
    m = res["mix"]["dur"].get(k, (float("nan"), 0))
    print("%-32s %8d %10.1f %10.1f %7.2f" % (k, s[1], s[0], m[0], m[0] / s[0] if s[0] else 0))


def g(mode, k, c):
    return res[mode]["pmc"].get(k, {}).get(c)


def fmt(v):
    return "%12s" % "-" if v is None else "%12.3g" % v


print("\n## counters per launch (serial | mix), derived")
rows = [("wave-parked share  SQ_WAIT_ANY / SQ_WAVE_CYCLES", lambda m, k: ratio(g(m, k, "SQ_WAIT_ANY"), g(m, k, "SQ_WAVE_CYCLES"))),
        ("issue-stall share  SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES", lambda m, k: ratio(g(m, k, "SQ_WAIT_INST_ANY"), g(m, k, "SQ_WAVE_CYCLES"))),
        ("VMEM read instructions", lambda m, k: g(m, k, "SQ_INSTS_VMEM_RD")),
        ("vector-cache accesses  TCP_TOTAL_CACHE_ACCESSES", lambda m, k: g(m, k, "TCP_TOTAL_CACHE_ACCESSES_sum")),
        ("TCP -> L2 read requests", lambda m, k: g(m, k, "TCP_TCC_READ_REQ_sum")),
        ("TCP pending-stall cycles per access", lambda m, k: ratio(g(m, k, "TCP_PENDING_STALL_CYCLES_sum"), g(m, k, "TCP_TOTAL_CACHE_ACCESSES_sum"))),
        ("TCP data-return stall cycles per access", lambda m, k: ratio(g(m, k, "TCP_TCP_TA_DATA_STALL_CYCLES_sum"), g(m, k, "TCP_TOTAL_CACHE_ACCESSES_sum"))),
        ("L2 requests", lambda m, k: g(m, k, "TCC_REQ_sum")),
        ("L2 hit rate", lambda m, k: ratio(g(m, k, "TCC_HIT_sum"), (g(m, k, "TCC_HIT_sum") or 0) + (g(m, k, "TCC_MISS_sum") or 0))),
        ("L2 tag-stall cycles per request", lambda m, k: ratio(g(m, k, "TCC_TAG_STALL_sum"), g(m, k, "TCC_REQ_sum"))),
        ("fabric read requests  TCC_EA0_RDREQ", lambda m, k: g(m, k, "TCC_EA0_RDREQ_sum")),
        ("fabric read latency (cycles) = RDREQ_LEVEL / RDREQ", lambda m, k: ratio(g(m, k, "TCC_EA0_RDREQ_LEVEL_sum"), g(m, k, "TCC_EA0_RDREQ_sum"))),
        ("DRAM credit-stall cycles per fabric read", lambda m, k: ratio(g(m, k, "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum"), g(m, k, "TCC_EA0_RDREQ_sum")))]


def ratio(a, b):
    return None if a is None or not b else a / b


for k in kern[:8]:
    print("\n%s" % k)
    for name, fn in rows:
        print("  %-58s %s | %s" % (name, fmt(fn("serial", k)), fmt(fn("mix", k))))
