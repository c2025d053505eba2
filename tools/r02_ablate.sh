# usage: bash tools/r02_ablate.sh <tag>  -- GPU box: where does the overlapped step go?  default bench (no CPU baseline, no
# extract leg) with phases of the k-hop sampler switched off (FGNN_KHOP_ABLATE: 1 swap simulation, 2 dedup insert,
# 4 CSR write-back) and with khop2's batch order dropped.  Results with a switch on are WRONG by design: timing only.
tag=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
out=gpurun_out/${tag}_ablate.txt; : > $out
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
run no_writeback FGNN_KHOP_ABLATE=4
run no_swap_sim FGNN_KHOP_ABLATE=1
run no_insert FGNN_KHOP_ABLATE=2
run unordered FGNN_KHOP2_UNORDERED=1
run unordered_no_writeback FGNN_KHOP2_UNORDERED=1 FGNN_KHOP_ABLATE=4
run base_again FGNN_X=0
