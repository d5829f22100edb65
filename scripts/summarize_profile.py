#!/usr/bin/env python3
"""Condense rocprofv3 output (kernel stats + PMC counter passes) into small files for profiles/."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

out_dir, tag = sys.argv[1], sys.argv[2]
res = {}
stats = glob.glob(os.path.join(out_dir, "trace", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    rows = list(csv.DictReader(open(stats[0])))
    keep = []
    for r in rows[:40]:
        keep.append({k: r[k] for k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs")})
    with open(os.path.join(out_dir, f"{tag}_kernel_stats_top40.csv"), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(keep[0].keys()))
        w.writeheader()
        w.writerows(keep)
for name in ("fetch", "write"):
    files = glob.glob(os.path.join(out_dir, f"pmc_{name}", "**", "*counter_collection.csv"), recursive=True)
    if not files:
        continue
    agg = defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(files[0])):
        k = re.sub(r"\(.*", "", r["Kernel_Name"])[:80]
        agg[(k, r["Counter_Name"])][0] += 1
        agg[(k, r["Counter_Name"])][1] += float(r["Counter_Value"])
    res[name] = [
        {"kernel": k, "counter": c, "dispatches": n, "sum": v, "per_dispatch": v / n}
        for (k, c), (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:30]
    ]
with open(os.path.join(out_dir, f"{tag}_pmc_summary.json"), "w") as f:
    json.dump(res, f, indent=1)
print(json.dumps({k: v[:8] for k, v in res.items()}, indent=1)[:3000])
