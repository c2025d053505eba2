# link band of the trainer's one-launch extraction against a trainer that also TRAINS (bench.py --gpus 2: 1S+1T on one GPU, the span with a training step per batch)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06n
for w in 32 64 16 32; do
  SAMGRAPH_EXTRACT_LINK_WGS=$w timeout -k 10 500 python3 bench.py --gpus 2 --steps 60 --no-cpu-baseline --no-n1-point > gpurun_out/r06n/gpus2_$w.json 2> gpurun_out/r06n/gpus2_$w.err || exit 1
  python3 - <<P
import json
l=json.loads(open('gpurun_out/r06n/gpus2_$w.json').read().strip().splitlines()[-1])
e=l['epoch_time_s']
print('band=%2s  extract-only %.4f ms/batch   with training %.4f ms/batch (windows %s)' % ('$w', l['ms_per_step'], e['with_training']/151*1e3, [round(x,3) for x in e['training_windows_ms_per_step']]))
P
done 2>&1 | tee gpurun_out/r06n/summary.txt
