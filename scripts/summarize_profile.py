#!/usr/bin/env python3
"""Condense rocprofv3 output (kernel stats + separate PMC counter passes) into the small files kept under profiles/.

usage: summarize_profile.py <gpurun_out/prof_TAG> <TAG> [commit]
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

def template_args(name: str, kernel: str):
    """['float', '4', 'true', 'true', 'false'] for '...::bwd_pair_kernel<float, 4, true, true, false>(...)', or None if
    `kernel` is not the kernel of `name` (exact base name: 'bwd_pair_kernel' does not match 'bwd_pair_split_kernel')."""
    m = re.search(r"(?:^|[\s:])" + re.escape(kernel) + r"<([^<>]*)>", name)
    return [a.strip() for a in m.group(1).split(",")] if m else None


def arg_is(name, kernel, index, value, default=None):
    """True if the template argument `index` of `kernel` in `name` equals `value`.  Only the listed argument is looked at:
    arguments appended to a kernel's template list later on do not change the classification (the round-2 regexes
    matched the whole list, so the fifth argument that `bwd_pair_kernel` gained silently emptied `roofline.traffic`)."""
    args = template_args(name, kernel)
    if args is None:
        return False
    got = args[index] if index < len(args) else default
    return got == value


def plain(*subs):
    return lambda k: any(s in k for s in subs)


# bench.py region name -> (predicates identifying the main kernel family, predicates of helper kernels that run once
# per launch).  Template layouts (generated_spec/*.hip): bwd_edge_kernel<T, WPN, FUSED, GW, GY, FULL>,
# bwd_pair_kernel<T, WPN, FULL, GX, DUAL>, bwd_pair_split_kernel<T, FULL, GX, ATOM>, bwd_pair_ring_kernel<WPN, GX, ATOM>.
REGIONS = {
    "tp_fwd": ([plain("::fwd_kernel<", "tp_fwd_kernel")], []),
    "tp_bwd_edge": ([lambda k: arg_is(k, "bwd_edge_kernel", 2, "false"), plain("tp_bwd_edge_kernel"),
                     lambda k: arg_is(k, "bwd_pair_kernel", 3, "false"),
                     lambda k: arg_is(k, "bwd_pair_split_kernel", 2, "false"),
                     lambda k: arg_is(k, "bwd_pair_ring_kernel", 1, "false")],
                    [plain("spec_gy_reduce_kernel", "tp_ypart_reduce_kernel")]),
    "tp_bwd_x": ([plain("::bwd_x_kernel<", "tp_bwd_x_kernel")], []),
    "tp_bwd_fused": ([lambda k: arg_is(k, "bwd_edge_kernel", 2, "true"),
                      lambda k: arg_is(k, "bwd_pair_kernel", 3, "true"),
                      lambda k: arg_is(k, "bwd_pair_split_kernel", 2, "true"),
                      lambda k: arg_is(k, "bwd_pair_ring_kernel", 1, "true")],
                     # (the accumulator form's hipMemsetAsync of [N, dim_in1] -- a runtime fill kernel, 23 MB written in cfg-3 --
                     # has no name of its own and is not counted)
                     [plain("gx_rows_sum_kernel", "gx_acc_finish_kernel")]),
    "radial_mlp_fwd": ([lambda k: re.search(r"radial_mlp_fwd\w*_kernel", k) is not None],
                       [plain("radial_mlp_split_w1_fwd_kernel", "radial_mlp_split_w1_fwd_f16_kernel")]),
    "radial_mlp_bwd": ([lambda k: re.search(r"radial_mlp_bwd\w*_kernel", k) is not None],
                       [plain("radial_mlp_transpose_w1_kernel", "radial_mlp_split_w1_bwd_kernel")]),
    "node_linear": ([plain("node_linear_kernel", "node_linear_mfma_kernel", "node_linear_")], []),
    "node_fused": ([plain("node_fused_kernel")], []),
    "energy_head": ([plain("energy_head_fwd_kernel", "energy_head_bwd_kernel")], []),
    "gate": ([plain("gate_fwd_kernel", "gate_bwd_kernel")], []),
    "edge_embed_fwd": ([plain("edge_embed_fwd_kernel")], []),
    "edge_embed_bwd": ([plain("edge_embed_bwd_kernel")], []),
    "edge_vectors": ([plain("edge_vectors_fwd_kernel", "edge_vectors_bwd_kernel")], []),
}

CALIBRATION = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r5_pmc_calibration.json")


def load_calibration():
    try:
        return json.load(open(CALIBRATION))["factors"]
    except Exception:
        return {}


def factors_for(kernel: str, calib: dict):
    """(FETCH_SIZE factor, WRITE_SIZE factor) of a GPU kernel by its access pattern: the tensor-product kernels read and
    write 4 B per lane (lane = channel, one 256 B row segment per wave access, nontemporal result stores); the radial MLP,
    node and embedding kernels move 16 B per lane."""
    tp = any(s in kernel for s in ("::fwd_kernel<", "bwd_edge_kernel", "bwd_pair", "bwd_x_kernel", "gx_rows_sum", "gx_acc_finish", "tp_fwd_kernel",
                                    "tp_bwd", "spec_gy_reduce"))
    if tp:
        return (calib.get("fetch_4B_per_lane_rows") or 2.0, calib.get("write_4B_per_lane_rows_nontemporal") or 1.0)
    return (calib.get("fetch_16B_per_lane_stream") or 2.0, calib.get("write_16B_per_lane_lines_nontemporal") or 1.0)


def region_of(kernel_name: str):
    """bench.py region of a GPU kernel name ('main' or 'helper' role), or (None, None)."""
    k = kernel_name.replace("(anonymous namespace)::", "")
    for region, (main_pats, helper_pats) in REGIONS.items():
        if any(p(k) for p in main_pats):
            return region, "main"
        if any(p(k) for p in helper_pats):
            return region, "helper"
    return None, None


def main(out_dir, tag, commit=None):
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
    calib = load_calibration()
    for k, v in per.items():
        # gfx950: FETCH_SIZE counts 64 B per 128 B request on wide coalesced reads (x2, MI355X_MICROARCH.md); other access
        # widths and WRITE_SIZE as calibrated on known byte counts in this repo's own access patterns
        # (scripts/pmc_calibrate.py -> profiles/r5_pmc_calibration.json), 2.0 / 1.0 when no calibration file is present
        ff, wf = factors_for(k, calib)
        v["fetch_factor"], v["write_factor"] = ff, wf
        v["hbm_bytes_corrected"] = (ff * v["FETCH_SIZE_KB"] + wf * v["WRITE_SIZE_KB"]) * 1024.0

    bench = {}
    for region, (main_pats, helper_pats) in REGIONS.items():
        ks = [k for k in per if any(p(k) for p in main_pats + helper_pats)]
        if not ks:
            continue
        # bytes per *region launch*: dispatch-weighted mean over the kernels of the region's main kernel family, plus
        # helper kernels (reductions / prepasses) that run once per launch
        main = [k for k in ks if any(p(k) for p in main_pats)]
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
                "gfx950 correction: hbm bytes = (fetch_factor*FETCH_SIZE + write_factor*WRITE_SIZE) * 1024 with the factors "
                "of profiles/r5_pmc_calibration.json per access pattern (2 / 1 of MI355X_MICROARCH.md when uncalibrated).",
        "calibration": calib,
        "kernels": dict(sorted(per.items())),
        "bench_kernels": bench,
        "commit": commit or os.environ.get("NQA_COMMIT", "n/a"),
    }
    with open(os.path.join(out_dir, f"{tag}_pmc_summary.json"), "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(bench, indent=1)[:3000])


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None)
