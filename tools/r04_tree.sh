#!/bin/bash
# prefix search trees: parity, then the twitter shape (config 4): A/B tree on / off, and the serial kernel timeline
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04h; mkdir -p $O
python -m pytest tests/test_hip_parity.py tests/test_full_size_properties.py -q -m gpu -x -k "weighted or tree or other_samplers or twitter or bitmap or replacement" > $O/tests.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/tests.log
[ $rc -ne 0 ] && { grep -n "Error\|assert\|error" $O/tests.log | head -20; exit $rc; }
bash tools/ab_variants.sh r04h_ab "base;FGNN_PREFIX_TREE=0" --rounds 3 --steps 53 --workload twitter --modes sample
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/pc3 -- python3 bench.py --workload twitter --steps 20 --warmup 3 --no-overlap --no-cpu-baseline --timed-only > $O/prof_serial_twitter.log 2>&1 || { tail -5 $O/prof_serial_twitter.log; exit 1; }
cp $(find $O/pc3 -name "*kernel_stats.csv") $O/bench_twitter_serial_kernel_stats.csv
python3 tools/chain_timeline.py $O/pc3 10 > $O/timeline_serial_twitter.txt 2>&1; rm -rf $O/pc3
cat $O/timeline_serial_twitter.txt
