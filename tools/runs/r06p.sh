cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06p
timeout -k 10 900 python -m pytest tests/test_engine_gpu.py -m gpu -x -q --timeout 400 -k "last_batch or arch5_multi or five_epochs or switcher or learns" > gpurun_out/r06p/pytest.log 2>&1; rc=$?; tail -6 gpurun_out/r06p/pytest.log; exit $rc
