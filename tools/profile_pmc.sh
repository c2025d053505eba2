#!/bin/bash
# PMC traffic passes (separate runs, as the MI355X guide prescribes) of the serial default bench; $1 = tag
tag=${1:-r04f}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$tag; mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -- python3 bench.py --steps 20 --warmup 3 --no-overlap --no-cpu-baseline --timed-only > $O/pmc_$c.log 2>&1 || { tail -5 $O/pmc_$c.log; exit 1; }
done
python3 tools/pmc_summary.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_FETCH_SIZE.log > $O/pmc_traffic.json 2> $O/pmc.err
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
python3 - <<PY
import json
d=json.load(open("$O/pmc_traffic.json"))
for k,v in d["per_kernel"].items(): print("%-48s %-20s %6.2f MB/batch" % (k, v["stage"], v["hbm_bytes_per_batch"]/1e6))
for s,v in d["per_stage"].items(): print(s, round(v["hbm_bytes_per_batch"]/1e6,2), "MB vs algorithmic", round(v["algorithmic_bytes_per_batch"]/1e6,2), "ratio", round(v["traffic_over_algorithmic"],2))
PY
