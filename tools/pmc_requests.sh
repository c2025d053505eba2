#!/bin/bash
# usage: bash tools/pmc_requests.sh <tag> [workloads...]  -- GPU box.  Fabric requests per batch of the sampler-side stage
# (sample + dedup + remap + cache split), per kernel, for bench.py's roofline_sample: ONE --pmc pass per workload
# (TCC_EA0_RDREQ_sum + TCC_EA0_WRREQ_sum: two of the four TCC slots; --kernel-trace only, as the guide prescribes) of the
# one-stream run (a counter pass serialises launches anyway).  The run's warm-up holds bench.py's random-read probe, so
# the same pass also says how many requests ONE probe read is.
tag=$1; shift
wls=${@:-papers100M twitter uk-2006-05}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$tag; mkdir -p $O
for wl in $wls; do
  timeout -k 10 500 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --kernel-trace --output-format csv -d $O/${wl}_req -- python3 bench.py --workload $wl --steps 20 --warmup 3 --windows 1 --no-overlap --no-cpu-baseline --timed-only > $O/${wl}_req.log 2>&1 || { tail -5 $O/${wl}_req.log; exit 1; }
  echo "$wl done"
done
python3 tools/pmc_requests_summary.py $O $wls > $O/pmc_requests.json 2> $O/pmc_requests.err; tail -3 $O/pmc_requests.err
python3 - <<P
import json
d = json.load(open("$O/pmc_requests.json"))
for wl, r in d["workloads"].items():
    print(wl, "sampler-side requests per batch: read %.0f write %.0f; one probe read = %.3f requests" % (r["sampler_side_per_batch"]["read"], r["sampler_side_per_batch"]["write"], r["probe_requests_per_read"]))
    for k, v in sorted(r["kernels"].items(), key=lambda kv: -kv[1]["read_per_batch"] - kv[1]["write_per_batch"]):
        print("   %-34s launches/batch %.2f  read %.0f  write %.0f" % (k, v["launches_per_batch"], v["read_per_batch"], v["write_per_batch"]))
P
for wl in $wls; do rm -rf $O/${wl}_req; done
