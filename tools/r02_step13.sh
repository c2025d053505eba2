tag=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
FGNN_BENCH_WATCHDOG=150 timeout -k 10 300 python3 bench.py --gpus 4 --steps 60 --warmup 6 --train-steps 20 --no-cpu-baseline > gpurun_out/${tag}_gpus4.json 2> gpurun_out/${tag}_gpus4.err; echo "gpus4 rc=$?"
tail -c 400 gpurun_out/${tag}_gpus4.err
python3 tools/show_bench.py gpurun_out/${tag}_gpus4.json | grep -E "value|ms_per_step|busy|edges_per_s|rows_per_s|GBps|with_training|training_steps|sample_plus|parallelism|samplers|trainers"
