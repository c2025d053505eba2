# usage: bash tools/r02_prof.sh <tag>   -- GPU box: serial + overlapped kernel timelines of the default bench
tag=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/pt1 -- python3 bench.py --steps 40 --warmup 5 --no-overlap --no-cpu-baseline --timed-only > gpurun_out/${tag}_prof_serial.log 2>&1
python3 tools/chain_timeline.py gpurun_out/pt1 20 > gpurun_out/${tag}_timeline_serial.txt 2>&1; rm -rf gpurun_out/pt1
cat gpurun_out/${tag}_timeline_serial.txt
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pt2 -- python3 bench.py --no-cpu-baseline --timed-only > gpurun_out/${tag}_prof_overlap.log 2>&1
python3 tools/overlap_stats.py gpurun_out/pt2 > gpurun_out/${tag}_overlap.txt 2>&1
cp $(find gpurun_out/pt2 -name "*kernel_stats.csv") gpurun_out/${tag}_kernel_stats.csv
cp $(find gpurun_out/pt2 -name "*kernel_trace.csv") gpurun_out/${tag}_kernel_trace.csv
rm -rf gpurun_out/pt2
tail -30 gpurun_out/${tag}_overlap.txt
python3 tools/chain_period.py gpurun_out/${tag}_kernel_trace.csv > gpurun_out/${tag}_chain_period.txt 2>&1; cat gpurun_out/${tag}_chain_period.txt
