# usage: bash tools/r02_step2.sh <tag>  -- GPU box: whole GPU suite on the new build (split last layer, batched insert kernel,
# bitmap seed ranking), then bench A/B over streams and the twitter / uk shapes
tag=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_pytest.log 2>&1; rc=$?
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
run default
run streams2 --streams-per-thread 2
run streams4 --streams-per-thread 4
FGNN_KHOP_SPLIT_L0=0 run fused
run twitter --workload twitter --steps 53
run uk --workload uk-2006-05 --steps 60
run default_again
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/pt1 -- python3 bench.py --steps 40 --warmup 5 --no-overlap --no-cpu-baseline --timed-only > gpurun_out/${tag}_prof_serial.log 2>&1
python3 tools/chain_timeline.py gpurun_out/pt1 20 > gpurun_out/${tag}_timeline_serial.txt 2>&1; rm -rf gpurun_out/pt1
cat gpurun_out/${tag}_timeline_serial.txt
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/pt1 -- python3 bench.py --workload twitter --steps 20 --warmup 3 --no-overlap --no-cpu-baseline --timed-only > gpurun_out/${tag}_prof_serial_tw.log 2>&1
python3 tools/chain_timeline.py gpurun_out/pt1 10 > gpurun_out/${tag}_timeline_serial_twitter.txt 2>&1; rm -rf gpurun_out/pt1
cat gpurun_out/${tag}_timeline_serial_twitter.txt
