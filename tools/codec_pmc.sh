#!/bin/bash
# PMC passes over the codec leg (per-kernel averages): SQ activity split, then HBM bytes.  GPU box; writes gpurun_out/codec_pmc.json
set -e
export MTTS_LEG_REPS=1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d /tmp/pmc1 -o p -- python3 $R/tools/codec_leg.py > /tmp/pmc1.log 2>&1 || { tail -5 /tmp/pmc1.log; exit 1; }
echo pass1 $(date +%T)
timeout -k 10 300 rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_INSTS_SALU SQ_ACTIVE_INST_SCA --output-format csv -d /tmp/pmc2 -o p -- python3 $R/tools/codec_leg.py > /tmp/pmc2.log 2>&1 || { tail -5 /tmp/pmc2.log; exit 1; }
echo pass2 $(date +%T)
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE WRITE_SIZE --output-format csv -d /tmp/pmc3 -o p -- python3 $R/tools/codec_leg.py > /tmp/pmc3.log 2>&1 || { tail -5 /tmp/pmc3.log; exit 1; }
echo pass3 $(date +%T)
python3 - <<'P'
import csv, glob, json, collections, os
out = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ('/tmp/pmc1', '/tmp/pmc2', '/tmp/pmc3'):
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            name = r['Kernel_Name'].split('(')[0].replace('void ', '').strip()
            key = name + ' grid=' + r.get('Grid_Size', '?')
            out[key][r['Counter_Name']].append(float(r['Counter_Value']))
res = {k: {c: {'n': len(v), 'avg': sum(v) / len(v)} for c, v in cs.items()} for k, cs in out.items()}
json.dump(res, open(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/codec_pmc.json', 'w'), indent=1)
print(len(res), 'kernel/grid groups')
P
