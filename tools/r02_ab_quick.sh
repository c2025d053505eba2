# usage: bash tools/r02_ab_quick.sh <tag> [pytest -k expression]  -- GPU box: parity subset, then the default bench three times
tag=$1; sel=${2:-"driver or pipeline or sampler or layered"}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_hip_parity.py tests/test_full_size_properties.py tests/test_coresidency_gpu.py -m gpu -x -q -k "$sel" > gpurun_out/${tag}_pytest.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -4 gpurun_out/${tag}_pytest.log
[ $rc -ne 0 ] && exit $rc
run() {
  name=$1; shift
  timeout -k 10 400 python3 bench.py --no-cpu-baseline --no-extract-leg "$@" > gpurun_out/ab_tmp.json 2> gpurun_out/ab_tmp.err || { echo "$name FAILED"; tail -5 gpurun_out/ab_tmp.err; return 0; }
  python3 - "$name" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/ab_tmp.json").read().strip().splitlines()[-1])
r = d["roofline"]; s = d.get("sample_stage") or {}
print("%-28s step %.4f ms  sample-stage %.4f ms  gather overlapped %.1f us serial %.1f us  edges/step %.0f" % (
    sys.argv[1], d["ms_per_step"], s.get("ms_per_step", -1), r["avg_launch_ms"] * 1e3, (r.get("serial") or {}).get("avg_launch_ms", -1) * 1e3, d["edges_per_step"]))
PY
}
run default | tee gpurun_out/${tag}_ab.txt
run default2 | tee -a gpurun_out/${tag}_ab.txt
run default3 | tee -a gpurun_out/${tag}_ab.txt
run twitter --workload twitter --steps 53 | tee -a gpurun_out/${tag}_ab.txt
