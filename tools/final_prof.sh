cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
# 1. default bench (with cpu baseline), plain
python3 bench.py > gpurun_out/r01c_bench_default.json 2> gpurun_out/r01c_bench_default.err
# 2. same command under rocprofv3 stats
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pc1 -- python3 bench.py --no-cpu-baseline --timed-only > gpurun_out/r01c_prof_default.log 2>&1
python3 tools/stats_summary.py gpurun_out/pc1 > gpurun_out/r01c_default_stats.md
cp $(find gpurun_out/pc1 -name "*kernel_stats.csv") gpurun_out/r01c_default_kernel_stats.csv; rm -rf gpurun_out/pc1
# 3. serial timeline
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/pc2 -- python3 bench.py --steps 40 --warmup 5 --no-overlap --no-cpu-baseline > gpurun_out/r01c_prof_serial.log 2>&1
python3 tools/chain_timeline.py gpurun_out/pc2 20 > gpurun_out/r01c_timeline_serial.txt 2>&1; rm -rf gpurun_out/pc2
# 4. phase probe
python3 tools/phase_probe.py > gpurun_out/r01c_phase_probe.txt 2>&1
# 5. PMC traffic of the gather (separate passes)
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_$c -- python3 bench.py --steps 20 --warmup 3 --no-overlap --no-cpu-baseline > gpurun_out/r01c_pmc_$c.log 2>&1
done
python3 tools/pmc_summary.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE gpurun_out/r01c_pmc_FETCH_SIZE.log > gpurun_out/r01c_pmc_traffic.json 2> gpurun_out/r01c_pmc.err
rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
tail -c 600 gpurun_out/r01c_bench_default.json; cat gpurun_out/r01c_timeline_serial.txt; head -12 gpurun_out/r01c_pmc_traffic.json; cat gpurun_out/r01c_pmc.err | tail -3
# 6. the other workloads (config 4 / config 5 sampler side)
python3 bench.py --workload twitter > gpurun_out/r01c_bench_twitter.json 2> gpurun_out/r01c_bench_twitter.err
python3 bench.py --workload uk-2006-05 > gpurun_out/r01c_bench_uk.json 2> gpurun_out/r01c_bench_uk.err
tail -c 300 gpurun_out/r01c_bench_twitter.json; tail -c 300 gpurun_out/r01c_bench_uk.json
