# usage: bash tools/r02_run.sh <tag> [pytest|nopytest] [bench args...]   -- GPU box helper: tests, then the two bench modes
tag=$1; shift
what=$1; shift
mkdir -p gpurun_out
if [ "$what" = pytest ]; then
  timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_pytest.log 2>&1; rc=$?
  echo "pytest rc=$rc"; tail -5 gpurun_out/${tag}_pytest.log
  [ $rc -ne 0 ] && exit $rc
fi
timeout -k 10 400 python bench.py --steps 60 --warmup 10 "$@" > gpurun_out/${tag}_bench1.json 2> gpurun_out/${tag}_bench1.err; echo "bench1 rc=$?"
tail -c 1500 gpurun_out/${tag}_bench1.err
timeout -k 10 500 python bench.py --gpus 2 --steps 60 --warmup 10 "$@" > gpurun_out/${tag}_bench2.json 2> gpurun_out/${tag}_bench2.err; echo "bench2 rc=$?"
tail -c 1500 gpurun_out/${tag}_bench2.err
