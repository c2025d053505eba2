#!/bin/bash
# A/B env settings with the default bench on one box: prints ms/step, overlapped gather fraction
for cfg in "$@"; do
  env $cfg python3 bench.py --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('[$cfg]', '%.4f' % j['ms_per_step'], 'gather frac %.3f' % j['roofline']['frac'], 'avg gather us %.1f' % (j['roofline']['avg_launch_ms']*1e3))"
done
