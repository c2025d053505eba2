cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for c in FETCH_SIZE TCC_HIT_sum TCC_MISS_sum; do
timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_$c -- python3 bench.py --workload uk-2006-05 --steps 12 --warmup 3 --no-overlap --no-cpu-baseline --timed-only > gpurun_out/pmc_uk_$c.log 2>&1 || { echo "$c failed"; tail -3 gpurun_out/pmc_uk_$c.log; continue; }
python3 - $c <<'PY'
import csv, glob, sys
from collections import defaultdict
c = sys.argv[1]
f = glob.glob("gpurun_out/pmc_%s/**/*counter_collection.csv" % c, recursive=True)[0]
acc = defaultdict(list)
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] != c: continue
    n = r["Kernel_Name"]
    if "random_walk_topk" in n or "cache_split" in n or "ht_insert" in n:
        acc[n.split("(")[0][-40:]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    v.sort()
    print(c, k, "launches", len(v), "min %.0f median %.0f max %.0f" % (v[0], v[len(v)//2], v[-1]))
PY
rm -rf gpurun_out/pmc_$c
done
