cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06l
timeout -k 10 600 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "two_halves" > gpurun_out/r06l/pytest.log 2>&1; rc=$?; tail -15 gpurun_out/r06l/pytest.log; exit $rc
