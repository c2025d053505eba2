# usage: bash tools/ab_variants.sh <tag> "<variants>" [ab_variants.py args]: batch-driver parity under each variant, then the A/B
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ulimit -c 0
tag=$1; variants=$2; shift 2
mkdir -p gpurun_out/$tag
IFS=';' read -ra VS <<< "$variants"
for v in "${VS[@]}"; do
  [ "$v" = "base" ] && continue
  case "$v" in *UNORDERED*|*ABLATE*|*GATHER*) continue;; esac
  envs=$(echo "$v" | tr ',' ' ')
  log=gpurun_out/$tag/pytest_$(echo $v | tr '=,' '__').log
  env $envs timeout -k 10 600 python -m pytest tests/test_hip_parity.py tests/test_coresidency_gpu.py -m gpu -x -q --kernel-lib prof -k "driver or pipeline or coresid or stream" > $log 2>&1
  rc=$?; echo "parity [$v] rc=$rc $(tail -1 $log)"
  [ $rc -ne 0 ] && { tail -30 $log; exit $rc; }
done
timeout -k 10 600 python3 -u tools/ab_variants.py --variants "$variants" --out gpurun_out/$tag/ab.json "$@" > gpurun_out/$tag/ab.txt 2> gpurun_out/$tag/ab.err
rc=$?; cat gpurun_out/$tag/ab.txt; [ $rc -ne 0 ] && { echo "ab rc=$rc"; tail -5 gpurun_out/$tag/ab.err; }
exit $rc
