#!/bin/bash
# end-to-end epoch times on the papers100M-shaped dataset (one GPU): arch1, arch3 pipelined, arch5 1S+1T, arch6
ds=${1:-/tmp/ds/papers}
[ -f $ds/indptr.bin ] || python3 tools/make_big_dataset.py $ds papers100M > /dev/null 2>&1
export SAMGRAPH_EMPTY_FEAT=24
echo "== arch1"; python3 examples/train_graphsage.py --dataset-path $ds --arch arch1 --num-epoch 4 2>&1 | grep -E "Epoch 00[23]|test_result:epoch_time"
echo "== arch3 pipelined"; python3 examples/train_graphsage.py --dataset-path $ds --arch arch3 --pipeline --cache-percentage 1.0 --num-epoch 4 2>&1 | grep -E "Epoch 00[23]|test_result:epoch_time"
echo "== arch5 1S+1T"; python3 examples/multi_gpu/train_fgnn.py --dataset-path $ds --single-gpu --cache-percentage 1.0 --num-epoch 4 2>&1 | grep -E "Epoch 00[23]|test_result"
echo "== arch5 1S+1T HBM ring"; SAMGRAPH_DEVICE_RING_SLOTS=8 python3 examples/multi_gpu/train_fgnn.py --dataset-path $ds --single-gpu --cache-percentage 1.0 --num-epoch 4 2>&1 | grep -E "Epoch 00[23]|test_result"
echo "== arch5 no-train host / ring"
for s in 0 8; do SAMGRAPH_DEVICE_RING_SLOTS=$s python3 examples/multi_gpu/train_fgnn.py --dataset-path $ds --single-gpu --cache-percentage 1.0 --num-epoch 4 --no-train 2>&1 | grep -E "pipeline_train|sample_time"; done
echo "== arch6 1 worker"; python3 examples/sgnn/train_sgnn.py --arch arch6 --dataset-path $ds --cache-percentage 1.0 --num-epoch 4 2>&1 | grep -E "Epoch 00[23]|test_result:epoch_time"
