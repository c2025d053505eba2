# per-node records for the weighted draws (FGNN_PREFIX_REC=0: without), twitter shape: parity, interleaved A/B, counters
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06g
timeout -k 10 900 python -m pytest tests/test_hip_parity.py tests/test_full_size_properties.py tests/test_engine_gpu.py -m gpu -x -q -k "weighted or prefix or twitter" > gpurun_out/r06g/pytest.log 2>&1; rc=$?; tail -3 gpurun_out/r06g/pytest.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 900 python3 -u tools/ab_variants.py --variants "base;FGNN_PREFIX_REC=0" --workload twitter --rounds 5 --steps 40 --out gpurun_out/r06g/ab.json > gpurun_out/r06g/ab.txt 2> gpurun_out/r06g/ab.err; rc=$?; cat gpurun_out/r06g/ab.txt; [ $rc -ne 0 ] && { tail -5 gpurun_out/r06g/ab.err; exit $rc; }
bash tools/pmc_requests.sh r06g_req twitter
