# (the kernel this A/B measured -- fgnn_colsum, FGNN_TORCH_COLSUM=1 for torch -- was slower and is not in the tree: NOTES_rejected_experiments.md, round 6)
# fgnn_colsum (bias gradients + slice sums of the split weight-gradient GEMMs) against torch's reduce_kernel in the training step
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06u
timeout -k 10 600 python -m pytest tests/test_train_ops_gpu.py tests/test_hip_parity.py tests/test_examples_accuracy_gpu.py -m gpu -x -q -k "colsum or graphed_training or block_aggregate or learns or sage" > gpurun_out/r06u/pytest.log 2>&1; rc=$?; tail -3 gpurun_out/r06u/pytest.log; [ $rc -ne 0 ] && { grep -n "Error\|assert" gpurun_out/r06u/pytest.log | head; exit $rc; }
for v in 1 0 1 0; do
  if [ $v = 1 ]; then export FGNN_TORCH_COLSUM=1; else unset FGNN_TORCH_COLSUM; fi
  echo "== FGNN_TORCH_COLSUM=${FGNN_TORCH_COLSUM:-unset (fgnn_colsum)}"
  PYTORCH_TUNABLEOP_ENABLED=1 timeout -k 10 300 python3 tools/train_step_profile.py 2>&1 | grep -E "eager step|graph replay|graph:|reduce_kernel|colsum" | head -6
done 2>&1 | tee gpurun_out/r06u/ab.txt
