tag=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
FGNN_BENCH_WATCHDOG=400 timeout -k 10 500 python3 bench.py --gpus 4 --steps 60 --warmup 6 --train-steps 20 --no-cpu-baseline > gpurun_out/${tag}_gpus4.json 2> gpurun_out/${tag}_gpus4.err; echo "gpus4 rc=$?"
tail -c 600 gpurun_out/${tag}_gpus4.err
python3 tools/show_bench.py gpurun_out/${tag}_gpus4.json | grep -E "value|ms_per_step|busy|edges_per_s|rows_per_s|GBps|with_training|sample_plus|parallelism|samplers|trainers"
FGNN_BENCH_WATCHDOG=400 timeout -k 10 500 python3 bench.py --gpus 6 --samplers 2 --steps 60 --warmup 6 --train-steps 20 --no-cpu-baseline > gpurun_out/${tag}_gpus6.json 2> gpurun_out/${tag}_gpus6.err; echo "gpus6 rc=$?"
tail -c 600 gpurun_out/${tag}_gpus6.err
python3 tools/show_bench.py gpurun_out/${tag}_gpus6.json | grep -E "value|ms_per_step|busy|edges_per_s|rows_per_s|GBps|with_training|sample_plus|parallelism|samplers|trainers"
ls /dev/shm | head
