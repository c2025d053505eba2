import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    m = re.search(r"(?:fgnn|sam)::(?:\(anonymous namespace\)::)?(\w+)", r["Kernel_Name"]) or re.search(r"(__amd_rocclr_\w+)", r["Kernel_Name"])
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(1) if m else r["Kernel_Name"][:30], r.get("Stream_Id")))
rows.sort()
# the cached leg: gathers come in pairs per batch; find the region where two gather_rows16 follow a cache split
gaps = []
busy_end = rows[0][1]
for i in range(1, len(rows)):
    s, e, k, st = rows[i]
    if s - busy_end > 2_000_000:  # > 2 ms idle
        gaps.append((busy_end, s, (s - busy_end) / 1e6, rows[i - 1][2], k, i))
    busy_end = max(busy_end, e)
t0 = rows[0][0]
print("kernels", len(rows), "span s", (rows[-1][1] - t0) / 1e9)
for a, b, ms, kp, kn, i in gaps:
    print("idle %.1f ms at t=%.3f s  after %s before %s (kernel #%d)" % (ms, (a - t0) / 1e9, kp, kn, i))
