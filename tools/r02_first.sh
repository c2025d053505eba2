set -x
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02a_pytest.log 2>&1; echo "pytest rc=$?" 
tail -5 gpurun_out/r02a_pytest.log
timeout -k 10 400 python bench.py --steps 60 --warmup 10 > gpurun_out/r02a_bench1.json 2> gpurun_out/r02a_bench1.err; echo "bench1 rc=$?"
tail -c 3000 gpurun_out/r02a_bench1.err
timeout -k 10 500 python bench.py --gpus 2 --steps 40 --warmup 10 > gpurun_out/r02a_bench2.json 2> gpurun_out/r02a_bench2.err; echo "bench2 rc=$?"
tail -c 3000 gpurun_out/r02a_bench2.err
ls /dev/shm | head
