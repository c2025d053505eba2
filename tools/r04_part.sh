#!/bin/bash
# round 4: partitioned last fill -- parity (whole GPU suite), then interleaved A/B against the global-table insert
set -o pipefail
O=gpurun_out/r04b; mkdir -p $O
python -m pytest tests -q -m gpu -x > $O/gpu_tests.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 $O/gpu_tests.log
[ $rc -ne 0 ] && exit $rc
bash tools/ab_variants.sh r04b_ab "base;FGNN_HT_PARTITION=0" --rounds 5 --steps 151
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_s20.json 2> $O/bench_s20.err || echo "bench failed"
python3 -c "
import json
d=json.loads(open('$O/bench_s20.json').read().strip().splitlines()[-1])
print('ms/step', round(d['ms_per_step'],4), 'windows', [round(x,4) for x in d['windows']['ms_per_step']], 'host_enq', round(d['host_enqueue_ms_per_step'],4), 'gather frac', round(d['roofline']['frac'],3), 'stage', (d.get('sample_stage') or {}).get('ms_per_step'))
"
