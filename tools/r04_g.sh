#!/bin/bash
set -o pipefail
O=gpurun_out/r04g; mkdir -p $O
python -m pytest tests -q -m gpu -x > $O/gpu_tests.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/gpu_tests.log
[ $rc -ne 0 ] && { grep -n "Error\|assert" $O/gpu_tests.log | head; exit $rc; }
bash tools/r04_pmc.sh r04g_pmc
bash tools/ab_variants.sh r04g_ab "base;FGNN_HT_PARTITION=0" --rounds 5 --steps 151
for i in 1 2; do
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_s20_$i.json 2> $O/bench_s20_$i.err || echo "bench failed"
python3 -c "
import json
d=json.loads(open('$O/bench_s20_$i.json').read().strip().splitlines()[-1])
print('ms/step', round(d['ms_per_step'],4), 'windows', [round(x,4) for x in d['windows']['ms_per_step']], 'host_enq', round(d['host_enqueue_ms_per_step'],4), 'gather frac', round(d['roofline']['frac'],3), 'stage', (d.get('sample_stage') or {}).get('ms_per_step'), 'train', (d.get('train_leg') or {}).get('ms_per_step'), 'extract', (d.get('roofline_extract') or {}).get('ms_per_step'))
"
done
