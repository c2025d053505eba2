#!/bin/bash
# usage: bash tools/pmc_mix.sh <tag> [workloads...]  -- GPU box.  Why does the feature gather lose more than half of its
# rate beside the other batches' sampling chains on the D = 256 shapes?  Per workload: the default three-stream run
# ("mix") and the one-stream run ("serial"), each as a plain kernel trace (durations; do the kernels still overlap?) and
# as four separate --pmc passes (the guide's rule: counters in their own runs, --kernel-trace only):
#   SQ   wave cycles / wave-parked cycles / issue stalls / VMEM instructions
#   TCP  vector-cache accesses, requests to L2, cycles stalled on pending requests and on the data return path
#   TCC  L2 requests / hits / misses / tag stalls
#   EA   L2 -> fabric read requests, their accumulated in-flight level (= latency x rate), DRAM credit stalls
# tools/pmc_mix_summary.py condenses the directories into one table per workload.
tag=$1; shift
wls=${@:-twitter uk-2006-05}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$tag; mkdir -p $O
SQ="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD"
TCP="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"
TCC="TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum"
EA="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_BUSY_sum"
for wl in $wls; do
  for mode in mix serial; do
    extra=""; [ $mode = serial ] && extra="--no-overlap"
    args="--workload $wl --steps 24 --warmup 6 --windows 1 --no-cpu-baseline --timed-only $extra"
    timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/${wl}_${mode}_trace -- python3 bench.py $args > $O/${wl}_${mode}_trace.log 2>&1 || { tail -5 $O/${wl}_${mode}_trace.log; exit 1; }
    for set in SQ TCP TCC EA; do
      timeout -k 10 400 rocprofv3 --pmc ${!set} --kernel-trace --output-format csv -d $O/${wl}_${mode}_$set -- python3 bench.py $args > $O/${wl}_${mode}_$set.log 2>&1 || { tail -5 $O/${wl}_${mode}_$set.log; exit 1; }
      echo "$wl $mode $set done"
    done
  done
  python3 tools/pmc_mix_summary.py $O $wl > $O/${wl}_pmc_mix.txt 2> $O/${wl}_pmc_mix.err; tail -3 $O/${wl}_pmc_mix.err
  cat $O/${wl}_pmc_mix.txt
  rm -rf $O/${wl}_*_trace $O/${wl}_*_SQ $O/${wl}_*_TCP $O/${wl}_*_TCC $O/${wl}_*_EA
done
