#!/bin/bash
# round 6: the placement search's rule A/B in consecutive processes on one box: `first` = rounds 2-5 (the first candidate that is
# 3 % clear of its self-paired reference ends the search), `new` = round 6 (a merely clear candidate is remembered, up to 8 more
# are timed, the fastest clear pairing is kept; 7 % clear ends the search at once).  A 400^3 run in front stirs the device memory
# the way the driver's sequence (suite, smoke, bench) does.
# usage: bash tools/lab/placement_rule_ab.sh [pairs=4] [stir=1]
N=${1:-4}
if [ "${2:-1}" = "1" ]; then python bench.py --size 200 --steps 1 --warmup 0 --no-cpu --no-secondary > /dev/null 2>&1; fi
for i in $(seq 1 $N); do
  for rule in first 1; do
    STAN_PLACEMENT_TRACE=$rule python bench.py --steps 5 --warmup 2 --no-cpu --no-secondary 2> /tmp/pl.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']; s=c['placement_search']
print('rule %-5s value %.4e  spmv %.4f ms  frac %.3f  of stream %.3f  candidates %d kept %.4f slowest %.4f moved %s/%s' % ('$rule'.replace('1','new'), d['value'], r['avg_launch_ms'], r['frac'], r['frac_of_stream'], s['candidates_timed'], s['probe_ms_kept'], s['probe_ms_slowest'], s['vectors_moved_instead'], s['product_vectors_moved']))"
    grep "stan placement" /tmp/pl.err | cut -c18-140 | head -12 | sed 's/^/      /'
  done
done
