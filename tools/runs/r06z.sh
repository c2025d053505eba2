# the switcher example: its two GPU tests, then an A/B on the products shape (PinSAGE, 1S+2T on one GPU)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06z
timeout -k 10 600 python -m pytest tests/test_engine_gpu.py -m gpu -x -q -rs -k "switcher" > gpurun_out/r06z/pytest.log 2>&1; rc=$?
tail -8 gpurun_out/r06z/pytest.log
[ $rc -eq 0 ] || exit $rc
EX=examples/balance_switcher/train_switcher.py
timeout -k 10 300 python $EX --make-dataset products --dataset-path /tmp/sw_products --num-epoch 3 --single-gpu \
  --num-train-worker 2 --cache-percentage 0.2 --switch-cache-percentage 0.1 > gpurun_out/r06z/with_switcher.log 2>&1 || { tail -20 gpurun_out/r06z/with_switcher.log; exit 1; }
grep -E "test_result|Epoch 00[1-3]" gpurun_out/r06z/with_switcher.log
timeout -k 10 300 python $EX --dataset-path /tmp/sw_products --num-epoch 3 --single-gpu \
  --num-train-worker 2 --cache-percentage 0.2 --no-switcher > gpurun_out/r06z/no_switcher.log 2>&1 || { tail -20 gpurun_out/r06z/no_switcher.log; exit 1; }
grep -E "test_result|Epoch 00[1-3]" gpurun_out/r06z/no_switcher.log
