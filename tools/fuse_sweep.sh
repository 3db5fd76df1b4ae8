#!/bin/bash
# q/k/v epilogue inside the attention kernels (MTTS_FUSE_QKV_MAX = rows x pages up to which it is used) by batch size
for bc in "32 4096" "64 4096" "128 2048" "16 4096"; do
  set -- $bc
  for v in 1024 1000000; do
    MTTS_FUSE_QKV_MAX=$v python bench.py --steps 64 --warmup 16 --fake-context --no-codec --no-cpu-baseline --batch $1 --context $2 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('B=$1 ctx=$2 FUSE_QKV_MAX=$v', 'ms_per_step', round(d['ms_per_step'], 4), {k: round(x['avg_ms']*1e3, 2) for k, x in d['kernels'].items()})
"
  done
done
