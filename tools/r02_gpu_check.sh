# usage: bash tools/r02_gpu_check.sh <tag> [bench args]  -- GPU box: new tests first, then the serial timeline, the default bench, the whole GPU suite
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_coresidency_gpu.py tests/test_hip_parity.py::test_sanity_check_kernel -x -q > gpurun_out/${tag}_new.log 2>&1; rc=$?
echo "new tests rc=$rc"; tail -12 gpurun_out/${tag}_new.log
[ $rc -ne 0 ] && exit $rc
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/pt1 -- python3 bench.py --steps 40 --warmup 5 --no-overlap --no-cpu-baseline --timed-only > gpurun_out/${tag}_prof_serial.log 2>&1
python3 tools/chain_timeline.py gpurun_out/pt1 20 > gpurun_out/${tag}_timeline_serial.txt 2>&1; rm -rf gpurun_out/pt1
cat gpurun_out/${tag}_timeline_serial.txt
python3 bench.py --no-cpu-baseline "$@" > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
python3 tools/show_bench.py gpurun_out/${tag}_bench.json | grep -E '"value"|ms_per_step|frac|edges_per_s|hit_rate'
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/${tag}_pytest.log
