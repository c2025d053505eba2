#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04k; mkdir -p $O
python -m pytest tests/test_hip_parity.py -q -m gpu -x -k "graphed or fused_sage or example_layers or tall_linear or block_aggregate" > $O/tests.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/tests.log
[ $rc -ne 0 ] && { grep -n "Error\|assert\|error" $O/tests.log | head -30; tail -40 $O/tests.log; exit $rc; }
for flag in "" "--graphed-train"; do
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extract-leg $flag > $O/bench$flag.json 2> $O/bench.err || { echo "bench failed"; tail -20 $O/bench.err; }
python3 -c "
import json
d=json.loads(open('$O/bench$flag.json').read().strip().splitlines()[-1])
t=d.get('train_leg') or {}
print('[$flag] ms/step', round(d['ms_per_step'],4), 'train ms', t.get('ms_per_step'), t.get('host_ms_per_step'), t.get('step'), 'epoch', d.get('epoch_time_s',{}).get('with_training'))
"
done
SAGE_FUSED=1 ROWS=40 python3 tools/train_step_profile.py > $O/train_profile_fused.txt 2>&1; grep "^step" $O/train_profile_fused.txt
