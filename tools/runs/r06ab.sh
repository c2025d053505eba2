# open descriptors of the ranks of a pipeline run with every ring mapped (2 ranks, then 2S+3T on the one GPU)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06ab
timeout -k 10 400 python3 bench.py --gpus 2 --no-train-leg --no-cpu-baseline --no-n1-point > gpurun_out/r06ab/g2.json 2> gpurun_out/r06ab/g2.err || { tail -5 gpurun_out/r06ab/g2.err; exit 1; }
python3 -c "
import json
l=json.loads(open('gpurun_out/r06ab/g2.json').read().strip().splitlines()[-1]); p=l['pipeline']
print('gpus 2:', round(l['ms_per_step'],4), 'ms/batch', p['open_files'], 'sent_device', p['handoff']['sent_device'], 'degraded', p['handoff']['degraded'])"
timeout -k 10 500 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 5 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 5 --samplers 2 --no-train-leg --no-cpu-baseline --no-n1-point --workload products > gpurun_out/r06ab/g5.json 2> gpurun_out/r06ab/g5.err || { tail -5 gpurun_out/r06ab/g5.err; exit 1; }
python3 -c "
import json
l=[x for x in open('gpurun_out/r06ab/g5.json').read().strip().splitlines() if x.startswith('{')][-1]; l=json.loads(l); p=l['pipeline']
print('gpus 5 (2S+3T, products):', round(l['ms_per_step'],4), 'ms/batch', p['open_files'], 'sent_device', p['handoff']['sent_device'], 'degraded', p['handoff']['degraded'])"
