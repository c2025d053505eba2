# usage: bash tools/r02_split.sh <tag>  -- GPU box: A/B of the split last layer (FGNN_KHOP_SPLIT_L0): parity first, then bench
tag=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
FGNN_KHOP_SPLIT_L0=1 timeout -k 10 900 python -m pytest tests/test_hip_parity.py tests/test_full_size_properties.py -m gpu -x -q > gpurun_out/${tag}_split_pytest.log 2>&1; rc=$?
echo "split pytest rc=$rc"; tail -4 gpurun_out/${tag}_split_pytest.log
[ $rc -ne 0 ] && exit $rc
out=gpurun_out/${tag}_split_ab.txt; : > $out
run() {
  name=$1; shift
  env "$@" timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-extract-leg > gpurun_out/ab_tmp.json 2> gpurun_out/ab_tmp.err || { echo "$name FAILED" >> $out; tail -5 gpurun_out/ab_tmp.err >> $out; return 0; }
  python3 - "$name" >> $out <<'PY'
import json, sys
d = json.loads(open("gpurun_out/ab_tmp.json").read().strip().splitlines()[-1])
r = d["roofline"]; s = d.get("sample_stage") or {}
print("%-28s step %.4f ms  sample-stage %.4f ms  gather overlapped %.1f us serial %.1f us" % (
    sys.argv[1], d["ms_per_step"], s.get("ms_per_step", -1), r["avg_launch_ms"] * 1e3, (r.get("serial") or {}).get("avg_launch_ms", -1) * 1e3))
PY
  tail -1 $out
}
run base FGNN_X=0
run split FGNN_KHOP_SPLIT_L0=1
run base2 FGNN_X=0
run split2 FGNN_KHOP_SPLIT_L0=1
FGNN_KHOP_SPLIT_L0=1 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/pt1 -- python3 bench.py --steps 40 --warmup 5 --no-overlap --no-cpu-baseline --timed-only > gpurun_out/${tag}_prof_serial.log 2>&1
python3 tools/chain_timeline.py gpurun_out/pt1 20 > gpurun_out/${tag}_split_timeline_serial.txt 2>&1; rm -rf gpurun_out/pt1
cat gpurun_out/${tag}_split_timeline_serial.txt
