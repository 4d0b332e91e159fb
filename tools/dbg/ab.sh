#!/bin/bash
# alternate the round-2 tree (ab_r2/) and the current one: bench.py C3, events-off figure and the timed one
for i in 1 2 3; do
  for d in ab_r2 .; do
    (cd $d && python bench.py --no-cpu --no-side-leg --steps 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$d', 'ms', round(d['ms_per_step'],4), 'events_off', round(d['events_off']['ms_per_step'],4), 'sieve avg us', round(d['roofline']['avg_launch_us'],2))")
  done
done
