# GPU box: parity of the device-side chain hand-off (batch driver, co-residency stress, engine), then A/B against events
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=${1:-r03_flags}
mkdir -p gpurun_out/$tag
timeout -k 10 700 python -m pytest tests/test_hip_parity.py tests/test_coresidency_gpu.py tests/test_hip_golden_gpu.py -m gpu -x -q -k "driver or pipeline or batch or coresid or stream" > gpurun_out/$tag/pytest1.log 2>&1
rc=$?; echo "parity1 rc=$rc $(tail -1 gpurun_out/$tag/pytest1.log)"; [ $rc -ne 0 ] && { tail -30 gpurun_out/$tag/pytest1.log; exit $rc; }
timeout -k 10 700 python -m pytest tests/test_engine_gpu.py -m gpu -x -q -k "arch1_single_gpu or five_epochs or many_epochs or tiny_queue or arch5_multi" > gpurun_out/$tag/pytest2.log 2>&1
rc=$?; echo "parity2 rc=$rc $(tail -1 gpurun_out/$tag/pytest2.log)"; [ $rc -ne 0 ] && { tail -40 gpurun_out/$tag/pytest2.log; exit $rc; }
timeout -k 10 500 python3 tools/ab_variants.py --variants "base;FGNN_CHAIN_FLAGS=0;FGNN_KHOP2_UNORDERED=1" --out gpurun_out/$tag/ab3.json 2> gpurun_out/$tag/ab3.err | tee gpurun_out/$tag/ab3.txt
timeout -k 10 500 python3 tools/ab_variants.py --variants "base;FGNN_CHAIN_FLAGS=0" --streams 4 --rounds 3 --out gpurun_out/$tag/ab4.json 2> gpurun_out/$tag/ab4.err | tee gpurun_out/$tag/ab4.txt
timeout -k 10 500 python3 tools/ab_variants.py --variants "base;FGNN_CHAIN_FLAGS=0" --streams 2 --rounds 3 --out gpurun_out/$tag/ab2.json 2> gpurun_out/$tag/ab2.err | tee gpurun_out/$tag/ab2.txt
