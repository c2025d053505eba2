#!/bin/bash
# overlapped-only A/B of env settings on one box
ex='import sys,json
j=json.loads([l for l in sys.stdin if l.startswith("{")][-1])
print("%.4f" % j["ms_per_step"], "stage %.4f" % j["sample_stage"]["ms_per_step"])'
for cfg in "$@"; do
  o=$(env $cfg python3 bench.py --no-cpu-baseline 2>/dev/null | python3 -c "$ex")
  echo "[$cfg] $o"
done
