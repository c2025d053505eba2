# GPU box: the arch5 trainer stage ALONE (bench.py --gpus 2 --decoupled) for several grids of the host-source gather
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ulimit -c 0
tag=${1:-r03_trainer}
mkdir -p gpurun_out/$tag
: > gpurun_out/$tag/sweep.txt
for wgs in 0 256 128 64; do
  FGNN_GATHER_HOST_WGS=$wgs SAMGRAPH_DEVICE_RING_SLOTS=170 timeout -k 10 400 python3 bench.py --gpus 2 --decoupled --no-train-leg --no-cpu-baseline > gpurun_out/$tag/dec_$wgs.json 2> gpurun_out/$tag/dec_$wgs.err || { echo "wgs=$wgs FAILED"; tail -3 gpurun_out/$tag/dec_$wgs.err; continue; }
  python3 - $wgs gpurun_out/$tag/dec_$wgs.json <<'PY' | tee -a gpurun_out/$tag/sweep.txt
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
p = d["pipeline"]
print("host-gather workgroups %-5s trainer alone %.4f ms per batch (%.1f GB/s of miss rows), miss launch %.3f ms; sampler alone %.4f ms per batch" % (
    sys.argv[1] if sys.argv[1] != "0" else "1024", p["trainer_busy_s"] / d["steps"] * 1e3,
    p["miss"]["bytes_per_step"] / (p["trainer_busy_s"] / d["steps"]) / 1e9, p["miss"]["avg_launch_ms"], p["sampler_busy_s"] / d["steps"] * 1e3))
PY
done
