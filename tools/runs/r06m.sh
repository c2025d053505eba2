# smaller link bands, and what the launch's HBM band gets beside them (a GPU that only extracts)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06m
timeout -k 10 900 python3 -u tools/link_band_sweep.py --out gpurun_out/r06m/sweep.json > gpurun_out/r06m/sweep.txt 2>&1; rc=$?; cat gpurun_out/r06m/sweep.txt; exit $rc
