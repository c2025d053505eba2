#!/usr/bin/env python3
"""Samples the GPU's clocks and power from sysfs (no root needed) every `period_ms` while a command runs; every sample
carries CLOCK_MONOTONIC, the clock bench.py's per-batch stamps use (FGNN_BENCH_DUMP_STAMPS), so the two files line up.
usage: clock_trace.py out.txt period_ms -- command ..."""
import glob
import os
import re
import subprocess
import sys
import time

out, period = sys.argv[1], float(sys.argv[2]) * 1e-3
cmd = sys.argv[sys.argv.index("--") + 1:]
devs = sorted(glob.glob("/sys/class/drm/card*/device"))
dev = next((d for d in devs if os.path.exists(d + "/pp_dpm_sclk")), None)
files = {}
if dev:
    for name in ("pp_dpm_sclk", "pp_dpm_mclk", "pp_dpm_fclk", "pp_dpm_socclk"):
        if os.path.exists(dev + "/" + name):
            files[name] = dev + "/" + name
    for h in glob.glob(dev + "/hwmon/hwmon*"):
        for name in ("power1_average", "power1_input", "freq1_input", "freq2_input", "temp1_input"):
            if os.path.exists(h + "/" + name):
                files[name] = h + "/" + name
    if os.path.exists(dev + "/gpu_busy_percent"):
        files["gpu_busy_percent"] = dev + "/gpu_busy_percent"


def read(path):
    try:
        s = open(path).read()
    except OSError:
        return "?"
    if "*" in s:  # pp_dpm_*: the active level carries a star
        m = re.search(r"(\d+)\s*[Mm][Hh]z\s*\*", s)
        return m.group(1) if m else "?"
    return s.strip()


p = subprocess.Popen(cmd)
with open(out, "w") as f:
    f.write("# device %s\n# t_monotonic " % dev + " ".join(files) + "\n")
    while p.poll() is None:
        t = time.clock_gettime(time.CLOCK_MONOTONIC)
        f.write("%.6f " % t + " ".join(read(v) for v in files.values()) + "\n")
        time.sleep(period)
sys.exit(p.returncode)
