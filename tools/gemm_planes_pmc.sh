#!/bin/bash
# PMC of one decoder GEMM alone:  SHAPES=pw1 tools/gemm_planes_pmc.sh <tile_code>   (GPU box; prints per-kernel counter averages)
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; export ITERS=5
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_INSTS_MFMA SQ_INSTS_SALU SQ_WAVES SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  timeout -k 10 120 rocprofv3 --pmc $set --output-format csv -d /tmp/gp$i -o p -- python3 $R/tools/gemm_planes_bench.py "$@" > /tmp/gp$i.log 2>&1 || { echo "pass $i failed"; tail -3 /tmp/gp$i.log; continue; }
  echo "pass $i ok $(date +%T)"
done
python3 - <<'P'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('/tmp/gp*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'gemm_b3' in r['Kernel_Name']:
            acc[r['Kernel_Name'].split('(')[0][-40:]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()): print(f"   {c:34s} n={len(v)} avg={sum(v)/len(v):.4g}")
P
