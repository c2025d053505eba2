#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04i; mkdir -p $O
python -m pytest tests -q -m gpu -x > $O/gpu_tests.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/gpu_tests.log
[ $rc -ne 0 ] && { grep -n "Error\|assert\|error" $O/gpu_tests.log | head -20; exit $rc; }
for wl in twitter uk-2006-05; do
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/pc3 -- python3 bench.py --workload $wl --steps 20 --warmup 3 --no-overlap --no-cpu-baseline --timed-only > $O/prof_serial_$wl.log 2>&1 || { tail -5 $O/prof_serial_$wl.log; exit 1; }
cp $(find $O/pc3 -name "*kernel_stats.csv") $O/bench_${wl}_serial_kernel_stats.csv
python3 tools/chain_timeline.py $O/pc3 10 > $O/timeline_serial_$wl.txt 2>&1; rm -rf $O/pc3
cat $O/timeline_serial_$wl.txt
done
