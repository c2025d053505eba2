# smoke() and the driver's command twice, timed
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06o
python __graft_entry__.py smoke 2>&1 | tail -2
for i in 1 2; do
SECONDS=0; python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06o/bench_s20_$i.json 2> gpurun_out/r06o/bench_s20_$i.err || echo "bench failed"
echo "wall ${SECONDS} s"
python3 -c "
import json
d=json.loads(open('gpurun_out/r06o/bench_s20_$i.json').read().strip().splitlines()[-1])
r=d['roofline_extract']
print('ms/step', round(d['ms_per_step'],4), 'value %.3e' % d['value'], 'windows', [round(x,4) for x in d['windows']['ms_per_step']], 'host_enq', round(d['host_enqueue_ms_per_step'],4), 'gather frac', round(d['roofline']['frac'],3), 'stage', round(d['sample_stage']['ms_per_step'],4), 'rs', round(d['roofline_sample']['frac'],3), 'train', round(d['train_leg']['ms_per_step'],3), 'extract p1', round(r['ms_per_step'],4), r['hit_rate'], 'p3', round(r['variants']['presample_epoch_3']['ms_per_step'],4), 'cpu %.3e' % d['cpu_baseline']['value'])
"
done
