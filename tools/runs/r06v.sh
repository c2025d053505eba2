# the gather's launch shape again, now that the chain-first order overlaps more (>= 2 kernels running 91 % of the time)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06v
timeout -k 10 900 python3 -u tools/ab_variants.py --variants "base;FGNN_GATHER_UNROLL=2;FGNN_GATHER_UNROLL=2,FGNN_GATHER_WG_PER_CU=4;FGNN_GATHER_UNROLL=2,FGNN_GATHER_WG_PER_CU=5;FGNN_GATHER_UNROLL=2,FGNN_GATHER_WG_PER_CU=2;base" --rounds 7 --steps 151 --modes full --out gpurun_out/r06v/ab.json > gpurun_out/r06v/ab.txt 2> gpurun_out/r06v/ab.err; rc=$?; cat gpurun_out/r06v/ab.txt; [ $rc -ne 0 ] && tail -5 gpurun_out/r06v/ab.err; exit $rc
