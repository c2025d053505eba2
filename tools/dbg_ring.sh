export FGNN_BENCH_WATCHDOG=40
for cfg in "16 --decoupled" "80 " "80 --decoupled"; do
  set -- $cfg
  echo "=== ring slots $1 $2"
  SAMGRAPH_DEVICE_RING_SLOTS=$1 timeout -k 5 80 python bench.py --gpus 2 --workload small --steps 12 --warmup 4 --no-train-leg --empty-feat-bits 16 $2 > gpurun_out/dbg.json 2> gpurun_out/dbg.err; echo rc=$?
  grep -v "amdgpu.ids\|socket.cpp" gpurun_out/dbg.err | tail -12
  python tools/show_bench.py gpurun_out/dbg.json 2>/dev/null | grep -E "ms_per_step|busy"
done
