# the two stages of the pipeline each alone (--decoupled), contract ranking (presample_epoch 1) and the 3-epoch variant
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06h
for pe in 1 3; do
  SAMGRAPH_LOG_LEVEL=info timeout -k 10 500 python3 bench.py --gpus 2 --decoupled --no-train-leg --no-cpu-baseline --presample-epochs $pe > gpurun_out/r06h/decoupled_pe$pe.json 2> gpurun_out/r06h/decoupled_pe$pe.err || exit 1
  echo "== presample_epoch $pe"
  grep -E "sampler:|extraction thread" gpurun_out/r06h/decoupled_pe$pe.err | sed 's/^\[INFO\] [^ ]* //'
  python3 tools/show_bench.py gpurun_out/r06h/decoupled_pe$pe.json | grep -E '"ms_per_step"|busy|second_half|trainer_rows|band_GBps|hit_rate|"achieved"|"frac"' | head -20
done 2>&1 | tee gpurun_out/r06h/summary.txt
