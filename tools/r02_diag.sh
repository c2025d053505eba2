# usage: bash tools/r02_diag.sh <tag>  -- GPU box: where the step goes on the R-MAT graph: duplicate locality, in-kernel phase
# stamps, phase ablation of the k-hop sampler / khop2 order chain
tag=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python3 tools/dup_locality.py > gpurun_out/${tag}_dup_locality.txt 2>&1; echo "dup rc=$?"; tail -25 gpurun_out/${tag}_dup_locality.txt
timeout -k 10 300 python3 tools/phase_probe.py > gpurun_out/${tag}_phase_probe.txt 2>&1; echo "phase rc=$?"; tail -25 gpurun_out/${tag}_phase_probe.txt
bash tools/r02_ablate.sh $tag
cat gpurun_out/${tag}_ablate.txt
