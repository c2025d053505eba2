#!/bin/bash
# A/B a set of env settings on the same box: usage ab_bench.sh "VAR=val VAR2=val" "..." ; prints ms_per_step
# (overlapped, sampler-side stage, serial)
ex='import sys,json
j=json.loads([l for l in sys.stdin if l.startswith("{")][-1])
print("%.4f" % j["ms_per_step"], "stage %.4f" % j["sample_stage"]["ms_per_step"])'
for cfg in "$@"; do
  o=$(env $cfg python3 bench.py --no-cpu-baseline 2>/dev/null | python3 -c "$ex")
  s=$(env $cfg python3 bench.py --no-cpu-baseline --no-overlap 2>/dev/null | python3 -c "$ex")
  echo "[$cfg] overlapped $o | serial $s"
done
