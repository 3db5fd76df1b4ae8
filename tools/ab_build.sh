#!/bin/bash
# A/B of a COMPILE-TIME switch on the headline decode step: rebuilds libmtts.so on the box for each setting.
# usage: ab_build.sh "<flags A>" "<flags B>" [bench args]      e.g. ab_build.sh "" "-DMTTS_PK_WAVES=5"
A=$1; B=$2; shift 2
for v in "$A" "$B" "$A" "$B"; do
  MTTS_BUILD_FLAGS="$v" python moss-ttsd_amd/build.py --force > /dev/null 2>&1 || { echo "build failed for '$v'"; exit 1; }
  python bench.py --steps 96 --warmup 16 --fake-context --no-codec --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('flags [$v]', 'ms_per_step', round(d['ms_per_step'], 4), {k: round(x['avg_ms']*1e3, 2) for k, x in d['kernels'].items()})
"
done
MTTS_BUILD_FLAGS="" python moss-ttsd_amd/build.py --force > /dev/null 2>&1
