#!/bin/bash
# "ENV=..|<bench args>" sweep on one box: ms/step, sampler-side stage, overlapped gather fraction
for cfg in "$@"; do
  e=${cfg%%|*}; a=${cfg#*|}
  env $e python3 bench.py --no-cpu-baseline $a 2>/dev/null | python3 -c "
import sys,json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('[$cfg]', '%.4f' % j['ms_per_step'], 'stage %.4f' % j['sample_stage']['ms_per_step'], 'gather frac %.3f' % j['roofline']['frac'], 'serial gather us %.1f' % (j['roofline']['serial']['avg_launch_ms']*1e3))"
done
