#!/bin/bash
# decode step time by KV context at B=32 (fake context, sampling as in the headline bench)
for c in 640 1024 2048 3072 4096 6144 8192; do
  python bench.py --steps 64 --warmup 16 --fake-context --no-codec --no-cpu-baseline --context $c 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('ctx=$c', 'ms_per_step', round(d['ms_per_step'], 4), {k: round(x['avg_ms']*1e3, 2) for k, x in d['kernels'].items()})
"
done
