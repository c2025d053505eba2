#!/bin/bash
# final check of a build: whole GPU suite, smoke, then the driver's command twice
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r04z}; mkdir -p $O
python -m pytest tests -q -m gpu -x > $O/gpu_tests.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/gpu_tests.log
[ $rc -ne 0 ] && { grep -n "Error\|assert\|error" $O/gpu_tests.log | head -20; exit $rc; }
python __graft_entry__.py smoke 2>&1 | tail -2
for i in 1 2; do
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_s20_$i.json 2> $O/bench_s20_$i.err || echo "bench failed"
python3 -c "
import json
d=json.loads(open('$O/bench_s20_$i.json').read().strip().splitlines()[-1])
print('ms/step', round(d['ms_per_step'],4), 'value %.3e' % d['value'], 'windows', [round(x,4) for x in d['windows']['ms_per_step']], 'host_enq', round(d['host_enqueue_ms_per_step'],4), 'gather frac', round(d['roofline']['frac'],3), 'stage', round(d['sample_stage']['ms_per_step'],4), 'train', round(d['train_leg']['ms_per_step'],3), 'extract', round(d['roofline_extract']['ms_per_step'],4), 'cpu %.3e' % d['cpu_baseline']['value'], 'sdr', round(d['roofline']['traffic_over_algorithmic_per_kernel']['sample_dedup_remap']['traffic_over_algorithmic'],2))
"
done
