#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04l; mkdir -p $O
python -m pytest tests/test_hip_parity.py tests/test_engine_gpu.py -q -m gpu -x -k "cached or arch3 or arch5_multi" > $O/tests.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/tests.log
[ $rc -ne 0 ] && { tail -40 $O/tests.log; exit $rc; }
export FGNN_HIP_LIB=$PWD/fgnn-artifacts_amd/lib/libfgnn_hip_prof.so
for rep in 1 2; do for f in 1 0; do
FGNN_EXTRACT_FORK=$f python3 bench.py --gpus 1 --steps 64 --warmup 5 --windows 1 --no-cpu-baseline --no-train-leg > $O/bench_fork$f.json 2> $O/bench.err || { echo "bench failed"; tail -20 $O/bench.err; }
python3 -c "
import json
d=json.loads(open('$O/bench_fork$f.json').read().strip().splitlines()[-1])
e=d['roofline_extract']
print('fork=$f extract ms/step', round(e['ms_per_step'],4), 'miss GB/s', round(e['miss']['achieved'],1), 'frac', round(e['miss']['frac'],3), 'miss launch ms', round(e['miss']['avg_launch_ms'],4), 'cached launch ms', round(e['cached']['avg_launch_ms'],4), 'cached frac', round(e['cached']['frac'],3), 'hit', round(e['hit_rate'],3))
"
done; done
