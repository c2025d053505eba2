# one arch5 sampler in-process with nobody consuming, unprofiled (three runs), then the N = 1 bench's sampler-side stage
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06ac
for i in 1 2 3; do
  SAMGRAPH_LOG_LEVEL=info timeout -k 10 200 python3 tools/sampler_timeline.py > gpurun_out/r06ac/st_$i.log 2>&1 || { tail -5 gpurun_out/r06ac/st_$i.log; exit 1; }
  grep -E "sampler alone|per batch, from|sampler:" gpurun_out/r06ac/st_$i.log | sed 's/^\[INFO\] [^ ]* //'
done
timeout -k 10 300 python3 bench.py --steps 151 --warmup 5 --windows 3 --no-train-leg --no-cpu-baseline --no-extract-leg > gpurun_out/r06ac/bench.json 2> gpurun_out/r06ac/bench.err || { tail -5 gpurun_out/r06ac/bench.err; exit 1; }
python3 -c "
import json
d=json.loads(open('gpurun_out/r06ac/bench.json').read().strip().splitlines()[-1])
print('N=1 bench: step', round(d['ms_per_step'],4), 'sampler-side stage alone', d['sample_stage'])"
