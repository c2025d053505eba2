# every multi-process / multi-thread engine configuration of the GPU suite, 24 times each (rare races in the owed-tail /
# publisher logic would show here as a hang -> the run's own time limits)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06w
timeout -k 10 1100 bash tools/stress_engine.sh 24 2>&1 | tee gpurun_out/r06w/stress.txt
