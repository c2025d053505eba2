#!/bin/bash
# training step: GPU suite (the Python views changed), default bench line, rocprofv3 kernel summary of a run whose
# GPU time is dominated by the train leg
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04j; mkdir -p $O
python -m pytest tests -q -m gpu -x > $O/gpu_tests.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/gpu_tests.log
[ $rc -ne 0 ] && { grep -n "Error\|assert\|error" $O/gpu_tests.log | head -20; exit $rc; }
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_s20.json 2> $O/bench_s20.err || echo "bench failed"
python3 -c "
import json
d=json.loads(open('$O/bench_s20.json').read().strip().splitlines()[-1])
print('ms/step', round(d['ms_per_step'],4), 'host_enq', round(d['host_enqueue_ms_per_step'],4), 'stage', (d.get('sample_stage') or {}).get('ms_per_step'), 'train', d.get('train_leg'), 'epoch', d.get('epoch_time_s',{}).get('with_training'))
"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/pc1 -- python3 bench.py --steps 20 --warmup 5 --windows 1 --no-cpu-baseline --no-extract-leg --train-steps 200 > $O/prof_train.log 2>&1 || { tail -5 $O/prof_train.log; exit 1; }
python3 - <<PY
import csv, glob
f = glob.glob("$O/pc1/**/*kernel_stats.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
with open("$O/train_leg_kernel_stats.md", "w") as o:
    o.write("| kernel | calls | total ms | avg us | % of GPU time |\n|---|---|---|---|---|\n")
    for r in rows[:45]:
        o.write("| %s | %s | %.2f | %.1f | %.1f |\n" % (r["Name"][:110].replace("|", "/"), r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
print(open("$O/train_leg_kernel_stats.md").read()[:3500])
PY
rm -rf $O/pc1
grep -o '"train_leg": {[^}]*}' $O/prof_train.log | head -1
