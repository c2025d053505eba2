cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06a
timeout -k 10 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "fused_extraction or cached or gather" > gpurun_out/r06a/pytest_fused.log 2>&1; rc=$?; tail -5 gpurun_out/r06a/pytest_fused.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 900 python -m pytest tests/test_engine_gpu.py -m gpu -x -q -k "arch5 or arch3 or arch2 or switch" > gpurun_out/r06a/pytest_engine.log 2>&1; rc=$?; tail -5 gpurun_out/r06a/pytest_engine.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python3 -u tools/link_band_sweep.py --out gpurun_out/r06a/link_band_sweep.json > gpurun_out/r06a/link_band_sweep.txt 2>&1; rc=$?; tail -40 gpurun_out/r06a/link_band_sweep.txt; [ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python3 bench.py --steps 20 --warmup 5 --no-train-leg --no-cpu-baseline > gpurun_out/r06a/bench_quick.json 2> gpurun_out/r06a/bench_quick.err; rc=$?; tail -c 600 gpurun_out/r06a/bench_quick.json; exit $rc
