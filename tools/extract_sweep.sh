# GPU box: the N = 1 extract leg (features in host memory, 0.2 cache) for several grids of the host-source gather
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ulimit -c 0
tag=${1:-r03_extract}
mkdir -p gpurun_out/$tag
: > gpurun_out/$tag/sweep.txt
for wgs in 0 256 64 32 16; do
  FGNN_GATHER_HOST_WGS=$wgs timeout -k 10 300 python3 bench.py --steps 40 --windows 1 --no-cpu-baseline --no-train-leg > gpurun_out/$tag/bench_$wgs.json 2> gpurun_out/$tag/bench_$wgs.err || { echo "wgs=$wgs FAILED"; tail -3 gpurun_out/$tag/bench_$wgs.err; continue; }
  python3 - $wgs gpurun_out/$tag/bench_$wgs.json <<'PY' | tee -a gpurun_out/$tag/sweep.txt
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
e = d["roofline_extract"]
print("host-gather workgroups %-5s extract leg %.4f ms/step  miss %.1f GB/s over the region (frac %.2f), %.3f ms per miss launch, hit rate %.3f | headline %.4f ms/step" % (
    sys.argv[1] if sys.argv[1] != "0" else "1024", e["ms_per_step"], e["miss"]["achieved"], e["miss"]["frac"], e["miss"]["avg_launch_ms"], e["hit_rate"], d["ms_per_step"]))
PY
done
