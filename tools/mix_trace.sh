#!/bin/bash
# usage: bash tools/mix_trace.sh <tag> [workloads...] -- GPU box: plain kernel traces of the three-stream run and the
# one-stream run per workload (no counters: a --pmc pass serialises the launches), condensed by pmc_mix_summary.py
tag=$1; shift
wls=${@:-twitter uk-2006-05}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$tag; mkdir -p $O
for wl in $wls; do
  for mode in mix serial; do
    extra=""; [ $mode = serial ] && extra="--no-overlap"
    args="--workload $wl --steps 24 --warmup 6 --windows 1 --no-cpu-baseline --timed-only $extra"
    timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/${wl}_${mode}_trace -- python3 bench.py $args > $O/${wl}_${mode}_trace.log 2>&1 || { tail -5 $O/${wl}_${mode}_trace.log; exit 1; }
  done
  python3 tools/pmc_mix_summary.py $O $wl 2> $O/${wl}_mix.err | sed -n 1,20p > $O/${wl}_mix_trace.txt
  cat $O/${wl}_mix_trace.txt
  rm -rf $O/${wl}_*_trace
done
