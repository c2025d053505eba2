#!/bin/bash
# "threads,streams_per_thread" sweep on one box
for cfg in "$@"; do
  nt=${cfg%,*}; spt=${cfg#*,}
  python3 bench.py --no-cpu-baseline --host-threads $nt --streams-per-thread $spt 2>/dev/null | python3 -c "
import sys,json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('threads=$nt streams/thread=$spt', '%.4f' % j['ms_per_step'], 'stage %.4f' % j['sample_stage']['ms_per_step'], 'enqueue %.3f' % j['host_enqueue_ms_per_step'], 'overflow', j['overflow'], 'gather frac %.3f' % j['roofline']['frac'])"
done
