cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06j
timeout -k 10 1100 python -m pytest tests/test_engine_gpu.py tests/test_examples_accuracy_gpu.py -m gpu -x -v --timeout 400 > gpurun_out/r06j/pytest.log 2>&1; rc=$?; tail -5 gpurun_out/r06j/pytest.log; [ $rc -ne 0 ] && { grep -n "Error\|assert\|FAILED\|Timeout" gpurun_out/r06j/pytest.log | head -20; exit $rc; }
bash tools/runs/r06h.sh
