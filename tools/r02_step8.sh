tag=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_engine_gpu.py -m gpu -x -q -k "arch5 or ring or pipeline" > gpurun_out/${tag}_pytest.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -4 gpurun_out/${tag}_pytest.log
[ $rc -ne 0 ] && exit $rc
for q in 4 8; do
echo "== GPU_MAX_HW_QUEUES=$q"
GPU_MAX_HW_QUEUES=$q SAMGRAPH_LOG_LEVEL=info timeout -k 10 500 python3 tools/sampler_timeline.py 2>&1 | grep -E "sampler alone|sampler:"
GPU_MAX_HW_QUEUES=$q SAMGRAPH_LOG_LEVEL=info SAMGRAPH_DEVICE_RING_SLOTS=170 timeout -k 10 500 python3 bench.py --gpus 2 --decoupled --no-train-leg --no-cpu-baseline > gpurun_out/${tag}_decoupled_q$q.json 2> gpurun_out/${tag}_decoupled.err; echo "decoupled rc=$?"
grep -E "sampler:" gpurun_out/${tag}_decoupled.err
python3 tools/show_bench.py gpurun_out/${tag}_decoupled_q$q.json | grep -E "busy|sampler_side"
GPU_MAX_HW_QUEUES=$q timeout -k 10 400 python3 bench.py --no-cpu-baseline --no-extract-leg > gpurun_out/ab_tmp.json 2> gpurun_out/ab_tmp.err; python3 tools/show_bench.py gpurun_out/ab_tmp.json | grep -E '"ms_per_step"' | head -2
done
GPU_MAX_HW_QUEUES=8 timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/pt1 -- python3 tools/sampler_timeline.py > gpurun_out/${tag}_sampler_prof.log 2>&1; echo "prof rc=$?"
grep -E "sampler alone" gpurun_out/${tag}_sampler_prof.log
python3 tools/overlap_timeline.py gpurun_out/pt1 2000 400 > gpurun_out/${tag}_sampler_window.txt 2>&1
python3 tools/overlap_stats.py gpurun_out/pt1 | tail -3
rm -rf gpurun_out/pt1
head -45 gpurun_out/${tag}_sampler_window.txt
