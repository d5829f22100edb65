#!/usr/bin/env python3
"""Condense rocprofv3 output (kernel stats + separate PMC counter passes) into the small files kept under profiles/.

usage: summarize_profile.py <gpurun_out/prof_TAG> <TAG>
writes <dir>/<TAG>_kernel_stats_top40.csv and <dir>/<TAG>_pmc_summary.json:
  "kernels":       per GPU kernel: dispatches, FETCH_SIZE_KB, WRITE_SIZE_KB (per dispatch) and corrected HBM bytes
  "bench_kernels": the same bytes keyed by bench.py's HIP-event region names (what `roofline.traffic` reads)
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

out_dir, tag = sys.argv[1], sys.argv[2]

# bench.py region name -> substrings identifying the GPU kernels launched inside that region
REGIONS = {  # region: (regexes of the main kernel family, regexes of helper kernels that run once per launch)
    "tp_fwd": ([r"::fwd_kernel<", r"tp_fwd_kernel"], []),
    "tp_bwd_edge": ([r"::bwd_edge_kernel<float, \d, false", r"tp_bwd_edge_kernel",
                     r"::bwd_pair_kernel<float, \d, (true|false), false>"],
                    [r"spec_gy_reduce_kernel", r"tp_ypart_reduce_kernel"]),
    "tp_bwd_x": ([r"::bwd_x_kernel<", r"tp_bwd_x_kernel"], []),
    "tp_bwd_fused": ([r"::bwd_edge_kernel<float, \d, true", r"::bwd_pair_kernel<float, \d, (true|false), true>"],
                     [r"gx_rows_sum_kernel"]),
    "radial_mlp_fwd": ([r"radial_mlp_fwd(_bf16x6)?(_bal)?_kernel"], [r"radial_mlp_split_w1_fwd_kernel"]),
    "radial_mlp_bwd": ([r"radial_mlp_bwd(_bf16x6)?_kernel"],
                       [r"radial_mlp_transpose_w1_kernel", r"radial_mlp_split_w1_bwd_kernel"]),
    "node_linear": ([r"node_linear_kernel", r"node_linear_mfma_kernel"], []),
    "gate": ([r"gate_fwd_kernel", r"gate_bwd_kernel"], []),
    "edge_embed_fwd": ([r"edge_embed_fwd_kernel"], []),
    "edge_embed_bwd": ([r"edge_embed_bwd_kernel"], []),
    "edge_vectors": ([r"edge_vectors_fwd_kernel", r"edge_vectors_bwd_kernel"], []),
}

for sub, suffix in (("trace", ""), ("trace_serial", "_serial")):
    stats = glob.glob(os.path.join(out_dir, sub, "**", "*kernel_stats.csv"), recursive=True)
    if stats:
        rows = list(csv.DictReader(open(stats[0])))
        keep = [{k: r[k] for k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs")}
                for r in rows[:40]]
        with open(os.path.join(out_dir, f"{tag}{suffix}_kernel_stats_top40.csv"), "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=list(keep[0].keys()))
            w.writeheader()
            w.writerows(keep)

per = defaultdict(lambda: {"dispatches": 0, "FETCH_SIZE_KB": 0.0, "WRITE_SIZE_KB": 0.0})
for name, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    files = glob.glob(os.path.join(out_dir, f"pmc_{name}", "**", "*counter_collection.csv"), recursive=True)
    if not files:
        continue
    n = defaultdict(int)
    for r in csv.DictReader(open(files[0])):
        if r["Counter_Name"] != counter:
            continue
        k = re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", ""))[:90]
        if "nqa::" not in k:
            continue
        n[k] += 1
        per[k][counter + "_KB"] += float(r["Counter_Value"])
    for k, c in n.items():
        per[k][counter + "_KB"] /= c
        per[k]["dispatches"] = c
for k, v in per.items():
    # gfx950: FETCH_SIZE counts 64 B per 128 B request on wide coalesced reads (x2); WRITE_SIZE is exact, both in KB
    v["hbm_bytes_corrected"] = (2.0 * v["FETCH_SIZE_KB"] + v["WRITE_SIZE_KB"]) * 1024.0

bench = {}
for region, (main_pats, helper_pats) in REGIONS.items():
    ks = [k for k in per if any(re.search(p, k) for p in main_pats + helper_pats)]
    if not ks:
        continue
    # bytes per *region launch*: dispatch-weighted mean over the kernels of the region's main kernel family, plus
    # helper kernels (reductions / prepasses) that run once per launch
    main = [k for k in ks if any(re.search(p, k) for p in main_pats)]
    helpers = [k for k in ks if k not in main]
    nd = sum(per[k]["dispatches"] for k in main)
    if nd == 0:
        continue
    b = sum(per[k]["hbm_bytes_corrected"] * per[k]["dispatches"] for k in main) / nd
    for k in helpers:
        b += per[k]["hbm_bytes_corrected"] * per[k]["dispatches"] / nd
    bench[region] = {"hbm_bytes_per_launch": b, "dispatches": nd, "kernels": sorted(ks)}

res = {
    "note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE collected in separate passes over bench.py's cfg-3 workload "
            "(scripts/profile.sh). Units: KB per dispatch, averaged over that kernel's dispatches (all layers). "
            "gfx950 correction per MI355X_MICROARCH.md: hbm bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024.",
    "kernels": dict(sorted(per.items())),
    "bench_kernels": bench,
}
with open(os.path.join(out_dir, f"{tag}_pmc_summary.json"), "w") as f:
    json.dump(res, f, indent=1)
print(json.dumps(bench, indent=1)[:3000])
