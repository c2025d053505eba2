# the train leg without a host wait per step
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06t
for i in 1 2; do
timeout -k 10 500 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extract-leg > gpurun_out/r06t/bench_$i.json 2> gpurun_out/r06t/bench_$i.err || { tail -5 gpurun_out/r06t/bench_$i.err; exit 1; }
python3 - <<P
import json
l=json.loads(open('gpurun_out/r06t/bench_$i.json').read().strip().splitlines()[-1])
t=l['train_leg']; print('step', round(l['ms_per_step'],4), 'train leg', round(t['ms_per_step'],4), {k: round(v,4) for k,v in t['host_ms_per_step'].items()}, t['step'], 'regions', t['timed_regions_run'], 'epoch', l['epoch_time_s']['with_training'])
P
done
timeout -k 10 600 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "graphed_training" > gpurun_out/r06t/pytest.log 2>&1; tail -2 gpurun_out/r06t/pytest.log
