# shared-GPU extract leg (bench.py roofline_extract, presample_epoch 1) against the link band's shape -- profiling build
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06b
for v in "64 8" "64 4" "32 4" "16 4" "8 4" "32 8" "16 8" "128 4"; do
  set -- $v
  FGNN_FUSED_LINK_WGS=$1 FGNN_FUSED_LINK_UNROLL=$2 timeout -k 10 300 python3 bench.py --kernel-lib prof --steps 64 --warmup 5 --windows 1 --no-train-leg --no-cpu-baseline --presample-variants 3 > gpurun_out/r06b/bench_$1_$2.json 2> gpurun_out/r06b/bench_$1_$2.err || exit 1
  python3 - <<P
import json
l=json.loads(open('gpurun_out/r06b/bench_$1_$2.json').read().strip().splitlines()[-1])
r=l['roofline_extract']; v=r['variants']['presample_epoch_3']
print('band=%3s unroll=%s  presample1: %.4f ms/batch (link band %.3f ms, hbm band %.3f ms)   presample3: %.4f (%.3f, %.3f)   full path %.4f' % ('$1','$2', r['ms_per_step'], r['miss']['band_ms'], r['cached']['band_ms'], v['ms_per_step'], v['miss']['band_ms'], v['cached']['band_ms'], l['ms_per_step']))
P
done 2>&1 | tee gpurun_out/r06b/summary.txt
