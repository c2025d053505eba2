#!/bin/bash
# A/B a set of env settings on the same box: usage ab_bench.sh "VAR=val VAR2=val" "..." ; prints ms_per_step (overlapped, serial)
for cfg in "$@"; do
  o=$(env $cfg python3 bench.py --no-cpu-baseline 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*')
  s=$(env $cfg python3 bench.py --no-cpu-baseline --no-overlap 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*')
  echo "[$cfg] overlapped $o | serial $s"
done
