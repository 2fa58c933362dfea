#!/bin/bash
# round 6 (diagnosis): what does the placement search see in consecutive processes on one box, and what would going on past
# the first clear candidate have found?  STAN_PLACEMENT_TRACE=all times every candidate the bounds allow, then decides as usual.
# usage: bash tools/lab/placement_all_candidates.sh [runs=5]
N=${1:-5}
for i in $(seq 1 $N); do
  echo "== process $i"
  STAN_PLACEMENT_TRACE=all python bench.py --steps 3 --warmup 1 --no-cpu --no-secondary 2> /tmp/pl_$i.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']
print('   value %.4e  spmv %.4f ms  frac %.3f  of stream %.3f  search %s' % (d['value'], r['avg_launch_ms'], r['frac'], r['frac_of_stream'], c['placement_search']))"
  grep "stan placement" /tmp/pl_$i.err | cut -c1-160
done
