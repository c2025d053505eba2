tag=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_engine_gpu.py -m gpu -x -q > gpurun_out/${tag}_pytest.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -6 gpurun_out/${tag}_pytest.log
[ $rc -ne 0 ] && exit $rc
SAMGRAPH_LOG_LEVEL=info timeout -k 10 500 python3 tools/sampler_timeline.py > gpurun_out/${tag}_sampler_alone.txt 2>&1; echo "rc=$?"
grep -E "sampler alone|sampler:" gpurun_out/${tag}_sampler_alone.txt
SAMGRAPH_SAMPLER_SLOTS=3 SAMGRAPH_LOG_LEVEL=info timeout -k 10 500 python3 tools/sampler_timeline.py 2>&1 | grep -E "sampler alone|sampler:"
SAMGRAPH_SAMPLER_SLOTS=8 SAMGRAPH_SAMPLER_STREAMS=4 SAMGRAPH_LOG_LEVEL=info timeout -k 10 500 python3 tools/sampler_timeline.py 2>&1 | grep -E "sampler alone|sampler:"
timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/pt1 -- python3 tools/sampler_timeline.py > gpurun_out/${tag}_sampler_prof.log 2>&1; echo "prof rc=$?"
grep -E "sampler alone" gpurun_out/${tag}_sampler_prof.log
python3 tools/overlap_timeline.py gpurun_out/pt1 2000 400 > gpurun_out/${tag}_sampler_window.txt 2>&1
python3 tools/overlap_stats.py gpurun_out/pt1 > gpurun_out/${tag}_sampler_overlap.txt 2>&1
rm -rf gpurun_out/pt1
head -50 gpurun_out/${tag}_sampler_window.txt
tail -4 gpurun_out/${tag}_sampler_overlap.txt
SAMGRAPH_LOG_LEVEL=info SAMGRAPH_DEVICE_RING_SLOTS=170 timeout -k 10 500 python3 bench.py --gpus 2 --decoupled --no-train-leg --no-cpu-baseline > gpurun_out/${tag}_decoupled.json 2> gpurun_out/${tag}_decoupled.err; echo "decoupled rc=$?"
grep -E "sampler:|extraction thread" gpurun_out/${tag}_decoupled.err
python3 tools/show_bench.py gpurun_out/${tag}_decoupled.json | grep -E "value|ms_per_step|busy|edges_per_s|rows_per_s|GBps"
timeout -k 10 500 python3 bench.py --gpus 2 --no-cpu-baseline > gpurun_out/${tag}_gpus2.json 2> gpurun_out/${tag}_gpus2.err; echo "gpus2 rc=$?"
python3 tools/show_bench.py gpurun_out/${tag}_gpus2.json | grep -E "value|ms_per_step|busy|edges_per_s|rows_per_s|GBps|with_training|sample_plus"
