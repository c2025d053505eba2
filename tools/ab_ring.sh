#!/bin/bash
# arch5 (1 sampler + 1 trainer process sharing the GPU) epoch time with the host ring vs the HBM message ring
# usage: tools/ab_ring.sh <dataset dir made by tools/make_big_dataset.py or --make-dataset> [extra train_fgnn.py args]
ds=$1; shift
for slots in 0 8 0 8; do
  echo "== SAMGRAPH_DEVICE_RING_SLOTS=$slots"
  SAMGRAPH_DEVICE_RING_SLOTS=$slots python3 examples/multi_gpu/train_fgnn.py --dataset-path $ds --single-gpu \
    --num-sample-worker 1 --num-train-worker 1 --num-epoch 4 "$@" 2>&1 | grep -E "test_result|Epoch 00[23]"
done
