# the arch5 sampler PROCESS alone (--decoupled: it fills the ring with nobody draining) over batch buffers x streams,
# with the chain-first order: SAMGRAPH_SAMPLER_STREAMS / _SLOTS (default 3 streams, 9 buffers)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06aa
for v in "3 6" "2 4" "2 6" "3 9" "4 8" "4 12" "6 12" "3 6"; do
  set -- $v
  SAMGRAPH_SAMPLER_STREAMS=$1 SAMGRAPH_SAMPLER_SLOTS=$2 timeout -k 10 300 python3 bench.py --gpus 2 --decoupled --no-train-leg --no-cpu-baseline > gpurun_out/r06aa/dec_$1_$2.json 2> gpurun_out/r06aa/dec_$1_$2.err || { tail -5 gpurun_out/r06aa/dec_$1_$2.err; exit 1; }
  python3 - <<P
import json
l=json.loads(open('gpurun_out/r06aa/dec_$1_$2.json').read().strip().splitlines()[-1])
p=l['pipeline']
print('streams=$1 buffers=%-2s  sampler alone %.4f ms/batch   trainer alone (second half) %.4f' % ('$2', p['sampler_loop_ms_per_batch'], p['consumed_second_half_ms_per_batch']), flush=True)
P
done 2>&1 | tee gpurun_out/r06aa/summary.txt
