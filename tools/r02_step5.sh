tag=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
SAMGRAPH_LOG_LEVEL=info timeout -k 10 500 python3 tools/sampler_timeline.py > gpurun_out/${tag}_sampler_alone.txt 2>&1; echo "rc=$?"
grep -E "sampler alone|sampler:" gpurun_out/${tag}_sampler_alone.txt
timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/pt1 -- python3 tools/sampler_timeline.py > gpurun_out/${tag}_sampler_prof.log 2>&1; echo "prof rc=$?"
grep -E "sampler alone" gpurun_out/${tag}_sampler_prof.log
python3 tools/overlap_timeline.py gpurun_out/pt1 2000 600 > gpurun_out/${tag}_sampler_window.txt 2>&1
python3 tools/overlap_stats.py gpurun_out/pt1 > gpurun_out/${tag}_sampler_overlap.txt 2>&1
rm -rf gpurun_out/pt1
head -80 gpurun_out/${tag}_sampler_window.txt
tail -20 gpurun_out/${tag}_sampler_overlap.txt
