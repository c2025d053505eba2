# GPU box: A/B of the start-order tickets (FGNN_SCAN_TICKETS=0 -> tile = blockIdx.x): serial timeline + default bench
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for T in 0 1; do
  export FGNN_SCAN_TICKETS=$T
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/pt$T -- python3 bench.py --steps 40 --warmup 5 --no-overlap --no-cpu-baseline --timed-only > gpurun_out/ab_t${T}_serial.log 2>&1
  python3 tools/chain_timeline.py gpurun_out/pt$T 20 > gpurun_out/ab_t${T}_timeline.txt 2>&1; rm -rf gpurun_out/pt$T
  echo "== tickets=$T"; cat gpurun_out/ab_t${T}_timeline.txt
  python3 bench.py --no-cpu-baseline --no-extract-leg > gpurun_out/ab_t${T}_bench.json 2> gpurun_out/ab_t${T}_bench.err
  python3 tools/show_bench.py gpurun_out/ab_t${T}_bench.json | grep -E '"value"|ms_per_step|frac|edges_per_s'
done
