# usage: bash tools/profile_stages.sh <tag>  -- GPU box: the two stages of the N >= 2 pipeline each ALONE (bench.py --decoupled: the
# sampler process fills the ring, then the trainer process drains it; both ranks on the one GPU of the box, never at the
# same time), with the engine's own host-side split (log level info), and one arch5 sampler under rocprofv3
tag=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
out=gpurun_out/${tag}_pipeline_stages.txt; : > $out
SAMGRAPH_LOG_LEVEL=info SAMGRAPH_DEVICE_RING_SLOTS=170 timeout -k 10 500 python3 bench.py --gpus 2 --decoupled --no-train-leg --no-cpu-baseline > gpurun_out/${tag}_decoupled.json 2> gpurun_out/${tag}_decoupled.err; echo "decoupled rc=$?"
echo "== bench.py --gpus 2 --decoupled --no-train-leg (SAMGRAPH_DEVICE_RING_SLOTS=170, SAMGRAPH_LOG_LEVEL=info): 151 batches" >> $out
grep -E "sampler:|extraction thread" gpurun_out/${tag}_decoupled.err | sed 's/^\[INFO\] [^ ]* //' >> $out
python3 tools/show_bench.py gpurun_out/${tag}_decoupled.json | grep -E "busy|sampler_side|trainer_rows|band_GBps|hit_rate|bytes_per_step" >> $out
echo "== GPU_MAX_HW_QUEUES=8 rocprofv3 --kernel-trace -- python3 tools/sampler_timeline.py (one arch5 sampler in-process, nobody consuming)" >> $out
GPU_MAX_HW_QUEUES=8 timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/pt1 -- python3 tools/sampler_timeline.py > gpurun_out/${tag}_sampler_prof.log 2>&1; echo "prof rc=$?"
grep -E "sampler alone" gpurun_out/${tag}_sampler_prof.log >> $out
python3 tools/overlap_stats.py gpurun_out/pt1 | tail -3 >> $out
python3 tools/overlap_timeline.py gpurun_out/pt1 2000 300 >> $out 2>&1
rm -rf gpurun_out/pt1
cat $out
