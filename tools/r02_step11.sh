tag=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_hip_parity.py tests/test_full_size_properties.py -m gpu -x -q > gpurun_out/${tag}_pytest.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -4 gpurun_out/${tag}_pytest.log
[ $rc -ne 0 ] && exit $rc
for st in khop0; do
  timeout -k 10 400 python3 bench.py --no-cpu-baseline --no-extract-leg --sample-type $st > gpurun_out/ab_tmp.json 2> gpurun_out/ab_tmp.err
  echo "$st $(python3 tools/show_bench.py gpurun_out/ab_tmp.json | grep -E '\"value\"|\"ms_per_step\"|edges_per_step' | head -3 | tr -d '\n')"
done | tee gpurun_out/${tag}_khop0.txt
