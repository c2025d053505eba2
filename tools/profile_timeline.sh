#!/bin/bash
# serial (one batch at a time) kernel timeline + overlapped per-kernel stats of the default bench; $1 = tag
tag=${1:-r04d}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$tag; mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/pc2 -- python3 bench.py --steps 40 --warmup 5 --no-overlap --no-cpu-baseline --timed-only > $O/prof_serial.log 2>&1 || { tail -5 $O/prof_serial.log; exit 1; }
python3 tools/chain_timeline.py $O/pc2 20 > $O/timeline_serial.txt 2>&1; rm -rf $O/pc2
cat $O/timeline_serial.txt
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/pc1 -- python3 bench.py --no-cpu-baseline --timed-only > $O/prof_default.log 2>&1 || { tail -5 $O/prof_default.log; exit 1; }
python3 tools/stats_summary.py $O/pc1 > $O/default_stats.md
python3 tools/overlap_stats.py $O/pc1 > $O/overlap.txt 2>&1
rm -rf $O/pc1
head -25 $O/default_stats.md
grep -o '"ms_per_step": [0-9.]*' $O/prof_default.log | head -2
