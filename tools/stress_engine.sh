#!/bin/bash
# repeats every multi-process / multi-thread engine configuration of tests/test_engine_gpu.py; stops at the first failure
N=${1:-12}
run() {
  for i in $(seq 1 $N); do
    d=$(mktemp -d)
    timeout 200 python3 tests/engine_runner.py "$@" > gpurun_out/stress_out.txt 2> gpurun_out/stress_err.txt
    rc=$?
    rm -rf $d
    if [ $rc != 0 ]; then echo "FAILED: $* (iteration $i, rc=$rc)"; tail -5 gpurun_out/stress_out.txt; grep -v "amdgpu.ids" gpurun_out/stress_err.txt | tail -30; exit 1; fi
  done
  echo "ok x$N: $*"
}
shift
d=/tmp/stress_ds
run_cfg() { mode=$1; st=$2; shift 2; for i in $(seq 1 $N); do dd=$(mktemp -d); timeout 200 python3 tests/engine_runner.py $mode $st $dd "$@" > gpurun_out/stress_out.txt 2> gpurun_out/stress_err.txt; rc=$?; rm -rf $dd; if [ $rc != 0 ]; then echo "FAILED: $mode $st $* (iteration $i, rc=$rc)"; tail -5 gpurun_out/stress_out.txt; grep -v "amdgpu.ids" gpurun_out/stress_err.txt | tail -30; exit 1; fi; done; echo "ok x$N: $mode $st $*"; }
run_cfg arch5 khop2 1 1 0.25 pipeline
run_cfg arch5 khop2 1 1 0.0 inline
run_cfg arch5 khop2 2 1 0.25 pipeline
run_cfg arch5 weighted_khop_prefix 1 2 0.3 pipeline
run_cfg arch5 random_walk 2 2 0.2 inline
run_cfg arch5 khop1 1 1 0.2 pipeline
run_cfg switcher random_walk
run_cfg switcher khop2
run_cfg arch3 khop2 0.25 threads
run_cfg arch3 random_walk 0.0 threads
run_cfg arch4 weighted_khop_prefix 0.3 threads
