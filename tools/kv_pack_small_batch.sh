#!/bin/bash
# sealed KV pages on / off for small batches (latency-bound attention): ms per step, fake context
for bc in "1 2048" "2 2048" "4 2048" "1 4096" "4 4096" "8 4096" "16 4096"; do
  set -- $bc
  for v in 0 2; do
    MTTS_KV_PACK=$v python bench.py --steps 96 --warmup 16 --fake-context --no-codec --no-cpu-baseline --greedy --batch $1 --context $2 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('B=$1 ctx=$2 MTTS_KV_PACK=$v', 'ms_per_step', round(d['ms_per_step'], 4))
"
  done
done
