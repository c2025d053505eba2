# usage: bash tools/r02_step4.sh <tag>  -- GPU box: parity with the gather tail, default bench, sampler process host stats
tag=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_hip_parity.py tests/test_engine_gpu.py -m gpu -x -q -k "not example and not sgnn" > gpurun_out/${tag}_pytest.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -6 gpurun_out/${tag}_pytest.log
[ $rc -ne 0 ] && exit $rc
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
out=gpurun_out/${tag}_ab.txt; : > $out
run() {
  name=$1; shift
  timeout -k 10 400 python3 bench.py --no-cpu-baseline "$@" > gpurun_out/ab_tmp.json 2> gpurun_out/ab_tmp.err || { echo "$name FAILED" >> $out; tail -5 gpurun_out/ab_tmp.err >> $out; tail -3 $out; return 0; }
  python3 - "$name" >> $out <<'PY'
import json, sys
d = json.loads(open("gpurun_out/ab_tmp.json").read().strip().splitlines()[-1])
r = d["roofline"]; s = d.get("sample_stage") or {}; x = d.get("roofline_extract") or {}
print("%-28s step %.4f ms  sample-stage %.4f ms  gather overlapped %.1f us serial %.1f us  extract-leg %.4f ms  enqueue %.4f" % (
    sys.argv[1], d["ms_per_step"], s.get("ms_per_step", -1), r["avg_launch_ms"] * 1e3, (r.get("serial") or {}).get("avg_launch_ms", -1) * 1e3, x.get("ms_per_step", -1), d.get("host_enqueue_ms_per_step", -1)))
PY
  tail -1 $out
}
run default
run default2
SAMGRAPH_LOG_LEVEL=info SAMGRAPH_DEVICE_RING_SLOTS=170 timeout -k 10 500 python3 bench.py --gpus 2 --decoupled --no-train-leg --no-cpu-baseline > gpurun_out/${tag}_decoupled.json 2> gpurun_out/${tag}_decoupled.err; echo "decoupled rc=$?"
grep -E "sampler:|extraction thread" gpurun_out/${tag}_decoupled.err
python3 tools/show_bench.py gpurun_out/${tag}_decoupled.json | grep -E "value|ms_per_step|busy|edges_per_s|rows_per_s|GBps"
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/pt1 -- python3 bench.py --steps 40 --warmup 5 --no-overlap --no-cpu-baseline --timed-only > gpurun_out/${tag}_prof_serial.log 2>&1
python3 tools/chain_timeline.py gpurun_out/pt1 20 > gpurun_out/${tag}_timeline_serial.txt 2>&1; rm -rf gpurun_out/pt1
cat gpurun_out/${tag}_timeline_serial.txt
