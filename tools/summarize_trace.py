"""Summarise a rocprofv3 kernel-trace CSV: per kernel, calls / mean / median duration, and for the attention
passes the same restricted to full-context launches (grid of the decode step at the bench's KV length)."""
import csv, json, sys, collections, statistics
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    agg[name].append((dur, int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0)))
out = {}
for k, v in sorted(agg.items(), key=lambda kv: -sum(d for d, _ in kv[1])):
    d = [x for x, _ in v]
    e = {"calls": len(d), "mean_us": sum(d) / len(d), "median_us": statistics.median(d), "total_ms": sum(d) / 1e3}
    if k.startswith("attn_"):
        gmax = max(g for _, g in v)
        full = [x for x, g in v if g == gmax]
        e["full_context_calls"] = len(full)
        e["full_context_mean_us"] = sum(full) / len(full)
    out[k] = e
json.dump(out, open(sys.argv[2], "w"), indent=1)
for k, e in list(out.items())[:12]:
    print(k[:50], e)
