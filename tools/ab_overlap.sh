#!/bin/bash
# overlapped-only A/B of env settings on one box
for cfg in "$@"; do
  o=$(env $cfg python3 bench.py --no-cpu-baseline 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*')
  echo "[$cfg] $o"
done
