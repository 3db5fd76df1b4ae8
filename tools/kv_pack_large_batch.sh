for bc in "64 4096" "128 2048"; do
  set -- $bc
  for v in 0 1; do
    MTTS_KV_PACK=$v python bench.py --steps 64 --warmup 16 --fake-context --no-codec --no-cpu-baseline --batch $1 --context $2 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('B=$1 ctx=$2 MTTS_KV_PACK=$v', 'ms_per_step', round(d['ms_per_step'], 4), round(d['value']), 'ids/s')
"
  done
done
