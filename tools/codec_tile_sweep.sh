#!/bin/bash
# Per-shape times of every gemm_b3t_kernel variant: one rocprofv3 run of the codec leg per tile code, summarised by
# tools/rocpd_stats.py-style queries into gpurun_out/tile_sweep_w$WINDOWS.json (GPU box; WINDOWS windows per call, CODES = tile codes).
set -e
export WINDOWS=${WINDOWS:-8} MTTS_LEG_REPS=${MTTS_LEG_REPS:-3}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for t in ${CODES:-2222 2312 4221 3311 3411 4311 4411}; do
  MTTS_CODEC_TILE=$t rocprofv3 --kernel-trace -d /tmp/sweep_$t -o t -- python3 $R/tools/codec_leg_w.py > /tmp/sweep_$t.log 2>&1 || { tail -5 /tmp/sweep_$t.log; exit 1; }
  echo "done $t $(tail -1 /tmp/sweep_$t.log | cut -c1-40)"
done
python3 - <<'P'
import sqlite3, json, glob, os
out = {}
for d in sorted(glob.glob('/tmp/sweep_*/')):
    code = d.rstrip('/').split('_')[-1]
    db = sqlite3.connect(glob.glob(d + '/**/*.db', recursive=True)[0])
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    kd = [x for x in tabs if x.startswith('rocpd_kernel_dispatch')][0]
    ks = [x for x in tabs if x.startswith('rocpd_info_kernel_symbol')][0]
    rows = db.execute(f"select d.start, d.end-d.start, s.display_name from {kd} d join {ks} s on d.kernel_id=s.id order by d.start").fetchall()
    # the n-th b3t launch of a run is the same GEMM in every run: key by launch ordinal within the first call
    b3t = [r[1] for r in rows if 'gemm_b3t' in r[2]]
    out[code] = {"b3t_launches": len(b3t), "b3t_total_ms": sum(b3t) / 1e6, "all_ms": sum(r[1] for r in rows) / 1e6, "per_launch_us": [x / 1e3 for x in b3t]}
json.dump(out, open(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/tile_sweep_w' + os.environ.get('WINDOWS', '8') + '.json', 'w'))
for k, v in out.items(): print(k, v["b3t_launches"], round(v["b3t_total_ms"], 2), round(v["all_ms"], 2))
P
