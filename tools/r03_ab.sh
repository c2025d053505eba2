# usage: bash tools/r03_ab.sh <tag> "<variants>" [extra ab_variants.py args]   -- GPU box: parity of the batch driver under
# every non-base variant's environment (skipped for profiling-only switches), then the interleaved A/B
tag=$1; variants=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
IFS=';' read -ra VS <<< "$variants"
for v in "${VS[@]}"; do
  [ "$v" = "base" ] && continue
  case "$v" in *UNORDERED*|*ABLATE*) continue;; esac
  envs=$(echo "$v" | tr ',' ' ')
  env $envs timeout -k 10 600 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "driver or pipeline" > gpurun_out/$tag/pytest_$(echo $v | tr '=,' '__').log 2>&1
  rc=$?; echo "parity [$v] rc=$rc $(tail -1 gpurun_out/$tag/pytest_$(echo $v | tr '=,' '__').log)"
  [ $rc -ne 0 ] && exit $rc
done
timeout -k 10 900 python3 tools/ab_variants.py --variants "$variants" --out gpurun_out/$tag/ab.json "$@" 2> gpurun_out/$tag/ab.err | tee gpurun_out/$tag/ab.txt
