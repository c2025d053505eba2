# BASELINE configs 4 and 5 as PIPELINES (bench.py --gpus 2: 1S+1T sharing the box's one GPU): twitter-shaped GCN with
# weighted sampling, uk-shaped PinSAGE walks -- sampler process -> HBM ring -> trainer process, training span included
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06x
for wl in twitter uk-2006-05; do
  timeout -k 10 560 python3 bench.py --gpus 2 --workload $wl --steps 40 --no-cpu-baseline --no-n1-point > gpurun_out/r06x/gpus2_$wl.json 2> gpurun_out/r06x/gpus2_$wl.err; rc=$?
  echo "$wl rc=$rc"; [ $rc -ne 0 ] && { tail -c 1500 gpurun_out/r06x/gpus2_$wl.err; exit $rc; }
  python3 - <<P
import json
l=json.loads(open('gpurun_out/r06x/gpus2_$wl.json').read().strip().splitlines()[-1])
p=l['pipeline']; e=l['epoch_time_s']
print('$wl: %.3e edges/s, %.4f ms/batch (windows %s), hit rate %.3f, with training %s ms/batch, handoff %s' % (l['value'], l['ms_per_step'], [round(x,3) for x in l['windows']['ms_per_step']], p['hit_rate'], [round(x,3) for x in (e['training_windows_ms_per_step'] or [])], p['handoff']['transport']))
P
done 2>&1 | tee gpurun_out/r06x/summary.txt
