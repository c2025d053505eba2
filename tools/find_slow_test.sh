#!/bin/bash
# repeats the GPU suite until a run is abnormally slow, then shows which tests took the time
for i in $(seq 1 ${1:-6}); do
  s=$(date +%s)
  timeout 1500 python -m pytest tests -m gpu -x -q --durations=6 > gpurun_out/slow_$i.log 2>&1
  e=$(( $(date +%s) - s ))
  echo "run $i: ${e}s $(tail -1 gpurun_out/slow_$i.log)"
  if [ $e -gt 200 ]; then grep -A8 "slowest" gpurun_out/slow_$i.log; break; fi
done
