# GPU box: parity of the device-side chain hand-off, then ONE guarded A/B against events (stop at the first failure)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ulimit -c 0
tag=${1:-r03_flags3}
mkdir -p gpurun_out/$tag
timeout -k 10 600 python -m pytest tests/test_hip_parity.py tests/test_coresidency_gpu.py -m gpu -x -q -k "driver or pipeline or batch or coresid or stream" > gpurun_out/$tag/pytest1.log 2>&1
rc=$?; echo "parity1 rc=$rc $(tail -1 gpurun_out/$tag/pytest1.log)"; [ $rc -ne 0 ] && { tail -30 gpurun_out/$tag/pytest1.log; exit $rc; }
timeout -k 10 600 python -m pytest tests/test_full_size_properties.py -m gpu -x -q -k "full_size_invariants or products" > gpurun_out/$tag/pytest2.log 2>&1
rc=$?; echo "fullsize rc=$rc $(tail -1 gpurun_out/$tag/pytest2.log)"; [ $rc -ne 0 ] && { tail -30 gpurun_out/$tag/pytest2.log; exit $rc; }
timeout -k 10 400 python3 -u tools/ab_variants.py --variants "base;FGNN_CHAIN_FLAGS=0" --rounds 3 --out gpurun_out/$tag/ab3.json > gpurun_out/$tag/ab3.txt 2> gpurun_out/$tag/ab3.err
rc=$?; cat gpurun_out/$tag/ab3.txt; [ $rc -ne 0 ] && { echo "ab rc=$rc"; tail -5 gpurun_out/$tag/ab3.err; exit $rc; }
exit 0
