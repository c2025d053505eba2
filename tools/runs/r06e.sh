cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06e
timeout -k 10 1100 python -m pytest tests/test_examples_accuracy_gpu.py tests/test_engine_gpu.py tests/test_train_ops_gpu.py -m gpu -x -q -k "accuracy or learns or six_ranks or two_processes or xent or corrupted" > gpurun_out/r06e/pytest.log 2>&1; rc=$?; tail -30 gpurun_out/r06e/pytest.log; exit $rc
