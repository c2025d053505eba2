tag=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/pc2 -- python3 bench.py --steps 40 --warmup 5 --no-overlap --no-cpu-baseline --timed-only > gpurun_out/${tag}_prof_serial.log 2>&1 || exit 1
python3 tools/chain_timeline.py gpurun_out/pc2 20 > gpurun_out/${tag}_timeline_serial.txt 2>&1; rm -rf gpurun_out/pc2
cat gpurun_out/${tag}_timeline_serial.txt
for wl in twitter uk-2006-05; do
  timeout -k 10 500 python3 bench.py --workload $wl --steps 60 > gpurun_out/${tag}_bench_$wl.json 2> gpurun_out/${tag}_bench_$wl.err; echo "$wl rc=$?"
  python3 tools/show_bench.py gpurun_out/${tag}_bench_$wl.json | grep -E '"value"|ms_per_step|frac|edges_per_s|edges_per_step|input_nodes' | head -12
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pc3 -- python3 bench.py --workload $wl --steps 20 --warmup 3 --no-overlap --no-cpu-baseline --timed-only > gpurun_out/${tag}_prof_serial_$wl.log 2>&1 || exit 1
  cp $(find gpurun_out/pc3 -name "*kernel_stats.csv") gpurun_out/${tag}_bench_${wl}_serial_kernel_stats.csv
  python3 tools/chain_timeline.py gpurun_out/pc3 10 > gpurun_out/${tag}_timeline_serial_$wl.txt 2>&1; rm -rf gpurun_out/pc3
  tail -3 gpurun_out/${tag}_timeline_serial_$wl.txt
done
for st in khop0 khop1 weighted_khop weighted_khop_hash_dedup; do
  timeout -k 10 400 python3 bench.py --no-cpu-baseline --no-extract-leg --no-train-leg --sample-type $st > gpurun_out/ab_tmp.json 2> gpurun_out/ab_tmp.err
  echo "$st $(python3 tools/show_bench.py gpurun_out/ab_tmp.json | grep -E '\"value\"|\"ms_per_step\"|edges_per_step' | head -3 | tr -d '\n')"
done | tee gpurun_out/${tag}_sample_types.txt
