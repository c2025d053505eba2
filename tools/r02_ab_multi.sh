# usage: bash tools/r02_ab_multi.sh <tag> "<VAR=a>" "<VAR=b>" ...  -- GPU box: the default bench once per setting, twice around
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
: > gpurun_out/${tag}_ab.txt
for round in 1 2; do for v in "$@"; do
  env $v timeout -k 10 400 python3 bench.py --no-cpu-baseline --no-extract-leg --no-train-leg > gpurun_out/ab_tmp.json 2> gpurun_out/ab_tmp.err || { echo "$v FAILED"; continue; }
  python3 - "$v" <<'PY' | tee -a gpurun_out/${tag}_ab.txt
import json, sys
d = json.loads(open("gpurun_out/ab_tmp.json").read().strip().splitlines()[-1])
r = d["roofline"]; s = d.get("sample_stage") or {}
print("%-28s step %.4f ms  sample-stage %.4f ms  gather overlapped %.1f us serial %.1f us" % (
    sys.argv[1], d["ms_per_step"], s.get("ms_per_step", -1), r["avg_launch_ms"] * 1e3, (r.get("serial") or {}).get("avg_launch_ms", -1) * 1e3))
PY
done; done
