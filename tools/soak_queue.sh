# GPU box: soak of the sampler -> trainer ring on tiny queues (2-3 slots) with several samplers and trainers, many short
# epochs; every batch is checked bit for bit against the oracle by tests/engine_runner.py
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ulimit -c 0
mkdir -p gpurun_out/soak
fail=0
for rep in 1 2 3; do
  for cfg in "300000 2 2" "420000 2 3" "300000 1 3" "420000 3 2"; do
    set -- $cfg
    d=$(mktemp -d)
    FGNN_TEST_NUM_EPOCH=24 FGNN_TEST_NUM_TRAIN=1500 SAMGRAPH_MQ_BYTES=$1 timeout -k 10 300 python3 tests/engine_runner.py arch5 khop2 $d $2 $3 0.25 pipeline > gpurun_out/soak/run_${rep}_$1_$2_$3.log 2>&1
    rc=$?; rm -rf $d
    echo "rep $rep mq_bytes $1 ${2}S+${3}T rc=$rc $(grep -c 'checked' gpurun_out/soak/run_${rep}_$1_$2_$3.log) trainers reported"
    [ $rc -ne 0 ] && { fail=1; tail -5 gpurun_out/soak/run_${rep}_$1_$2_$3.log; }
  done
done
exit $fail
