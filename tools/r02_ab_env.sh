# usage: bash tools/r02_ab_env.sh <tag> <VAR=value> [rounds]  -- GPU box: default bench alternating without / with the variable
tag=$1; var=$2; rounds=${3:-3}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() {
  name=$1; shift
  env "$@" timeout -k 10 400 python3 bench.py --no-cpu-baseline --no-extract-leg > gpurun_out/ab_tmp.json 2> gpurun_out/ab_tmp.err || { echo "$name FAILED"; tail -5 gpurun_out/ab_tmp.err; return 0; }
  python3 - "$name" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/ab_tmp.json").read().strip().splitlines()[-1])
r = d["roofline"]; s = d.get("sample_stage") or {}
print("%-28s step %.4f ms  sample-stage %.4f ms  gather overlapped %.1f us serial %.1f us" % (
    sys.argv[1], d["ms_per_step"], s.get("ms_per_step", -1), r["avg_launch_ms"] * 1e3, (r.get("serial") or {}).get("avg_launch_ms", -1) * 1e3))
PY
}
: > gpurun_out/${tag}_ab.txt
for i in $(seq $rounds); do
  run "default" FGNN_X=0 | tee -a gpurun_out/${tag}_ab.txt
  run "$var" $var | tee -a gpurun_out/${tag}_ab.txt
done
