cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for S in 0 128; do
  if [ $S != 0 ]; then export FGNN_KHOP_S=$S; fi
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof9 -- python3 bench.py --steps 40 --warmup 5 --no-overlap --no-cpu-baseline > gpurun_out/prof9.log 2>&1; tail -1 gpurun_out/prof9.log | cut -c1-400
  python3 tools/chain_timeline.py gpurun_out/prof9 > gpurun_out/timeline_S$S.txt 2>&1; cat gpurun_out/timeline_S$S.txt; rm -rf gpurun_out/prof9
  python3 bench.py --no-cpu-baseline 2>&1 | tail -1 | cut -c1-330
done
