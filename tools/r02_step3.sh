# usage: bash tools/r02_step3.sh <tag>  -- GPU box: parity of the coalesced rank prefix, twitter A/B (bitmap ranking vs rocPRIM),
# the arch5 sampler process alone (decoupled pipeline)
tag=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_hip_parity.py tests/test_engine_gpu.py -m gpu -x -q -k "not example and not sgnn" > gpurun_out/${tag}_pytest.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -6 gpurun_out/${tag}_pytest.log
[ $rc -ne 0 ] && exit $rc
out=gpurun_out/${tag}_ab.txt; : > $out
run() {
  name=$1; shift
  timeout -k 10 400 python3 bench.py --no-cpu-baseline --no-extract-leg "$@" > gpurun_out/ab_tmp.json 2> gpurun_out/ab_tmp.err || { echo "$name FAILED" >> $out; tail -5 gpurun_out/ab_tmp.err >> $out; tail -3 $out; return 0; }
  python3 - "$name" >> $out <<'PY'
import json, sys
d = json.loads(open("gpurun_out/ab_tmp.json").read().strip().splitlines()[-1])
r = d["roofline"]; s = d.get("sample_stage") or {}
print("%-28s step %.4f ms  sample-stage %.4f ms  gather overlapped %.1f us serial %.1f us  edges/step %.0f" % (
    sys.argv[1], d["ms_per_step"], s.get("ms_per_step", -1), r["avg_launch_ms"] * 1e3, (r.get("serial") or {}).get("avg_launch_ms", -1) * 1e3, d["edges_per_step"]))
PY
  tail -1 $out
}
run twitter_rank --workload twitter --steps 53
FGNN_RANK_BITMAP=0 run twitter_rocprim --workload twitter --steps 53
run twitter_rank2 --workload twitter --steps 53
run khop1_rank --sample-type khop1
FGNN_RANK_BITMAP=0 run khop1_rocprim --sample-type khop1
SAMGRAPH_DEVICE_RING_SLOTS=170 timeout -k 10 500 python3 bench.py --gpus 2 --decoupled --no-train-leg --no-cpu-baseline > gpurun_out/${tag}_decoupled.json 2> gpurun_out/${tag}_decoupled.err; echo "decoupled rc=$?"
tail -c 600 gpurun_out/${tag}_decoupled.err
python3 tools/show_bench.py gpurun_out/${tag}_decoupled.json | grep -E "value|ms_per_step|busy|edges_per_s|rows_per_s|GBps"
