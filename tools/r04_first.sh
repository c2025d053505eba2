#!/bin/bash
# round 4, first GPU contact: GPU test-suite, then the driver's command and the long-window command
set -o pipefail
O=gpurun_out/r04a; mkdir -p $O
python -m pytest tests -q -m gpu > $O/gpu_tests.log 2>&1; echo "pytest rc=$?" | tee -a $O/gpu_tests.log
tail -5 $O/gpu_tests.log
for i in 1 2; do
  python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_s20_$i.json 2> $O/bench_s20_$i.err || echo "bench s20 failed"
  true
done
python3 bench.py --gpus 1 --steps 151 --warmup 10 --no-cpu-baseline --timed-only > $O/bench_s151.json 2> $O/bench_s151.err || echo "bench s151 failed"
python3 -c "
import json,sys
for f in ('bench_s20_1','bench_s20_2','bench_s151'):
    try:
        d=json.loads(open('$O/'+f+'.json').read().strip().splitlines()[-1])
        print(f, 'ms/step', round(d['ms_per_step'],4), 'windows', [round(x,4) for x in d['windows']['ms_per_step']], 'host_enq', round(d['host_enqueue_ms_per_step'],4), 'gather frac', round(d['roofline']['frac'],3), 'stage', (d.get('sample_stage') or {}).get('ms_per_step'))
    except Exception as e: print(f, 'ERR', e)
"
