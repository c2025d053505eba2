cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06q
timeout -k 10 600 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "driver_matches or gather or cached or overlapped" > gpurun_out/r06q/pytest.log 2>&1; rc=$?; tail -3 gpurun_out/r06q/pytest.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 500 python3 bench.py --no-cpu-baseline --no-train-leg --no-extract-leg > gpurun_out/r06q/bench.json 2> gpurun_out/r06q/bench.err || exit 1
python3 - <<P
import json
l=json.loads(open('gpurun_out/r06q/bench.json').read().strip().splitlines()[-1])
r=l['roofline']
print('ms/step', l['ms_per_step'], 'events', r['avg_launch_ms'], r['frac'], 'kernel clock', r['kernel_clock'], 'serial', r['serial']['frac'])
P
