#!/bin/bash
# A/B/A/B of one environment switch on the headline decode step (fake context: no ramp).  usage: ab_bench.sh VAR A B [bench args]
VAR=$1; A=$2; B=$3; shift 3
for i in 1 2; do
  for v in $A $B; do
    env $VAR=$v python bench.py --steps 96 --warmup 16 --fake-context --no-codec --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$VAR=$v', 'ms_per_step', round(d['ms_per_step'], 4), {k: round(x['avg_ms']*1e3, 2) for k, x in d['kernels'].items()})
"
  done
done
