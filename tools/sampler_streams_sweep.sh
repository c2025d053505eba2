# GPU box: the arch5 sampler stage ALONE (bench.py --gpus 2 --decoupled) for several stream / batch-buffer counts
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ulimit -c 0
tag=${1:-sampler_streams}
mkdir -p gpurun_out/$tag
: > gpurun_out/$tag/sweep.txt
for cfg in "3 6" "2 4" "2 6" "1 2" "4 8"; do
  set -- $cfg
  SAMGRAPH_SAMPLER_STREAMS=$1 SAMGRAPH_SAMPLER_SLOTS=$2 SAMGRAPH_DEVICE_RING_SLOTS=170 timeout -k 10 300 python3 bench.py --gpus 2 --decoupled --no-train-leg --no-cpu-baseline > gpurun_out/$tag/dec_$1_$2.json 2> gpurun_out/$tag/dec_$1_$2.err || { echo "cfg=$cfg FAILED"; tail -3 gpurun_out/$tag/dec_$1_$2.err; continue; }
  python3 - "$1 streams, $2 batch buffers" gpurun_out/$tag/dec_$1_$2.json <<'PY' | tee -a gpurun_out/$tag/sweep.txt
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
p = d["pipeline"]
print("%-28s sampler alone %.4f ms per batch = %.3e edges/s" % (sys.argv[1], p["sampler_busy_s"] / d["steps"] * 1e3, p["sampler_side_edges_per_s"]))
PY
done
