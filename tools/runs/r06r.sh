# accuracy actually reached by the examples on the learnable synthetic dataset (the margins of tests/test_examples_accuracy_gpu.py)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06r
D=/tmp/acc_ds/learn
python3 examples/train_graphsage.py --make-dataset learnable --dataset-path $D --arch arch1 --fanout 10 5 --batch-size 1000 --num-epoch 2 --num-hidden 64 --lr 0.01 --report-acc 25 2>&1 | grep -E "Acc|test_acc" | tr '\n' ' '; echo " <- arch1 fused"
python3 examples/train_graphsage.py --dataset-path $D --arch arch1 --fanout 10 5 --batch-size 1000 --num-epoch 2 --num-hidden 64 --lr 0.01 --report-acc 25 --op-by-op 2>&1 | grep -E "Acc|test_acc" | tr '\n' ' '; echo " <- arch1 op-by-op"
python3 examples/multi_gpu/train_fgnn.py --dataset-path $D --single-gpu --cache-percentage 0.2 --fanout 10 5 --batch-size 1000 --num-epoch 2 --num-hidden 64 --lr 0.01 --report-acc 25 2>&1 | grep -E "Acc|test_acc" | tr '\n' ' '; echo " <- arch5"
