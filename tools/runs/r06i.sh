# chain(k+1) before tail(k) in the native batch loop: parity (batch driver / overlapped soak / cached), then the default bench
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06i
timeout -k 10 900 python -m pytest tests/test_hip_parity.py tests/test_coresidency_gpu.py tests/test_full_size_properties.py -m gpu -x -q -k "driver or range or overlap or stream or cached or products or coresid" > gpurun_out/r06i/pytest.log 2>&1; rc=$?; tail -3 gpurun_out/r06i/pytest.log; [ $rc -ne 0 ] && { grep -n "Error\|assert" gpurun_out/r06i/pytest.log | head; exit $rc; }
for k in 1 2; do
timeout -k 10 500 python3 bench.py --no-cpu-baseline --no-train-leg > gpurun_out/r06i/bench_$k.json 2> gpurun_out/r06i/bench_$k.err || exit 1
python3 tools/show_bench.py gpurun_out/r06i/bench_$k.json | grep -E '"value"|"ms_per_step"|host_enqueue|sample_stage_ms|extract_leg' | head -8
done
timeout -k 10 500 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-train-leg > gpurun_out/r06i/bench_20.json 2> gpurun_out/r06i/bench_20.err || exit 1
python3 tools/show_bench.py gpurun_out/r06i/bench_20.json | grep -E '"value"|"ms_per_step"|host_enqueue|sample_stage_ms|extract_leg' | head -8
