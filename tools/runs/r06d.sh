cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06d
timeout -k 10 1100 python -m pytest tests -m gpu -x -q -rs > gpurun_out/r06d/pytest_gpu.log 2>&1; rc=$?; tail -15 gpurun_out/r06d/pytest_gpu.log; exit $rc
