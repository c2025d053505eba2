#!/usr/bin/env python3
"""The one-launch extraction's rates from KERNEL durations (rocprofv3 --kernel-trace), not from HIP events or device
stamps: launches of extract_fused_kernel in every process's trace under <dir>, their average duration, and the time at
least one of them was running (launches of up to four batches overlap: a launch's own duration is the link time of
~4 batches, the union is what the link was busy for).
usage: extract_kernel_rates.py <dir> <miss_bytes_per_launch> <hit_bytes_per_launch>   (bytes from the bench line)"""
import csv
import glob
import sys

d, miss_b, hit_b = sys.argv[1], float(sys.argv[2]), float(sys.argv[3])
print("| process trace | launches | avg kernel us | busy (>= 1 launch running) us per launch | link GB/s over busy time | of 64 GB/s | HBM-band GB/s if the band ran for the whole kernel (lower bound) |")
print("|---|---|---|---|---|---|---|")
for f in sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True)):
    iv = []
    for r in csv.DictReader(open(f)):
        if "extract_fused_kernel" in r["Kernel_Name"]:
            iv.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    if len(iv) < 8:
        continue
    iv.sort()
    iv = iv[len(iv) // 4:]  # the steady part
    avg = sum(e - s for s, e in iv) / len(iv) / 1e3
    busy, cur_s, cur_e = 0, iv[0][0], iv[0][1]
    for s, e in iv[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    per = busy / len(iv) / 1e3
    gbs = miss_b / (per * 1e-6) / 1e9
    print("| %s | %d | %.1f | %.1f | %.1f | %.2f | %.0f |" % (f.split("/")[-1][:40], len(iv), avg, per, gbs, gbs / 64.0,
                                                           hit_b / (avg * 1e-6) / 1e9))
