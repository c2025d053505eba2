# link band packed onto one XCD (FGNN_FUSED_XCD=1, profiling build) against the spread band: shared-GPU extract leg
# (the switch existed only in the build that measured it: no gain, removed -- profiles/NOTES_rejected_experiments.md, round 6)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06c
for v in "0 32 4" "1 32 4" "1 16 4" "1 64 4" "1 32 8" "1 64 8" "1 128 4" "0 16 4"; do
  set -- $v
  FGNN_FUSED_XCD=$1 FGNN_FUSED_LINK_WGS=$2 FGNN_FUSED_LINK_UNROLL=$3 timeout -k 10 300 python3 bench.py --kernel-lib prof --steps 64 --warmup 5 --windows 1 --no-train-leg --no-cpu-baseline --presample-variants 3 > gpurun_out/r06c/bench_$1_$2_$3.json 2> gpurun_out/r06c/bench_$1_$2_$3.err || exit 1
  python3 - <<P
import json
l=json.loads(open('gpurun_out/r06c/bench_$1_$2_$3.json').read().strip().splitlines()[-1])
r=l['roofline_extract']; v=r['variants']['presample_epoch_3']
print('xcd=%s band=%3s unroll=%s  presample1: %.4f ms/batch (link band %.3f ms, hbm band %.3f ms)   presample3: %.4f (%.3f, %.3f)   full path %.4f' % ('$1','$2','$3', r['ms_per_step'], r['miss']['band_ms'], r['cached']['band_ms'], v['ms_per_step'], v['miss']['band_ms'], v['cached']['band_ms'], l['ms_per_step']))
P
done 2>&1 | tee gpurun_out/r06c/summary.txt
FGNN_FUSED_XCD=1 timeout -k 10 300 python3 -u tools/link_band_sweep.py --bands 16,32,64,128 > gpurun_out/r06c/link_band_sweep_xcd.txt 2>&1; tail -20 gpurun_out/r06c/link_band_sweep_xcd.txt
