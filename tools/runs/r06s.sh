# the whole GPU suite twice more on the final build (flakiness check), then smoke
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06s
for i in 1 2; do
  timeout -k 10 560 python -m pytest tests -m gpu -x -q > gpurun_out/r06s/pytest_$i.log 2>&1; rc=$?; echo "run $i rc=$rc $(tail -1 gpurun_out/r06s/pytest_$i.log)"
  [ $rc -ne 0 ] && { grep -n "Error\|assert\|FAILED" gpurun_out/r06s/pytest_$i.log | head -20; exit $rc; }
done
python __graft_entry__.py smoke 2>&1 | tail -2
