"""Merge two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) into HBM bytes per launch per kernel.
usage: pmc_summary.py <dir_fetch> <dir_write> <out.json>
FETCH_SIZE / WRITE_SIZE are in KB; FETCH_SIZE is doubled (gfx950 tallies 128-B requests at 64 B,
/opt/skills/guides/MI355X_MICROARCH.md); bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024."""
import collections
import csv
import glob
import json
import os
import sys


def load(d, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"].split("(")[0].replace("void ", "").strip()].append(float(r["Counter_Value"]))
    return acc


fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
out = {}
for k, v in fetch.items():
    # the attention kernels: keep the full-context launches only (largest values; prefill launches are short)
    top = sorted(v)[len(v) // 2:] if k.startswith("attn_") else v
    w = write.get(k, [0.0])
    wtop = sorted(w)[len(w) // 2:] if k.startswith("attn_") else w
    f_kb, w_kb = sum(top) / len(top), sum(wtop) / len(wtop)
    out[k] = {"launches": len(v), "FETCH_SIZE_KB_raw": f_kb, "WRITE_SIZE_KB": w_kb,
              "hbm_bytes_per_launch": int((2 * f_kb + w_kb) * 1024)}
json.dump({"command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --output-format csv -- python3 bench.py --fake-context --layers 2 "
           "--prompt 40 --steps 2 --warmup 1 --profile-steps 1 --no-cpu-baseline --no-codec (two separate passes)",
           "note": "B=32 rows at KV length ~4095, one launch = one layer; see tools/pmc_summary.py for the correction",
           "kernels": out}, open(sys.argv[3], "w"), indent=1)
for k, e in sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"])[:10]:
    print(k[:60].ljust(60), e["launches"], round(e["hbm_bytes_per_launch"] / 1e6, 2), "MB")
