cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06y
timeout -k 10 900 python -m pytest tests/test_engine_gpu.py -m gpu -x -q --timeout 500 -k "weighted_and_walks or two_processes" > gpurun_out/r06y/pytest.log 2>&1; rc=$?; tail -6 gpurun_out/r06y/pytest.log; exit $rc
