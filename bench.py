#!/usr/bin/env python3
"""Benchmark of the NequIP message-passing hot path on MI355X.

Metric (BASELINE.json): atom-steps/s = atoms / wall time of one *energy + forces* evaluation of the model
(eval mode, neighbour list prebuilt and resident in HBM, float32 model / float64 positions).  Workload at N=1:
BASELINE config "10k-atom periodic water box, l_max=2, 64 features, 3 interaction layers" (SURVEY.md 8(d) cfg-3:
15^3 H2O = 10 125 atoms, r_max 4.5 A, parity=False, radial MLP 128x1).  With N>1 every rank evaluates its own
copy of the box (frames are independent; replicas, no data-path collective) and the value is the aggregate.

    python bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task statement) carrying `roofline` (dominant hand-written
kernel: algorithmic bytes / HIP-event duration on the launching stream) and `cpu_baseline` (the CPU oracle,
``oracle/``, timed on a bounded sample of the same workload on this host).
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA peak (MI355X_MICROARCH.md)
MFMA_F32_PEAK_TFLOPS = 157.3  # dense fp32 MFMA peak (v_mfma_f32_32x32x2_f32), same guide
# committed PMC summary (scripts/profile.sh): only used for `roofline.traffic` when the live measurement below is
# unavailable, and then labelled as static in `traffic_source`
PMC_SUMMARY = next((p for p in (os.path.join(ROOT, "profiles", f"r{r}_pmc_summary.json") for r in (3, 2)) if os.path.exists(p)),
                   os.path.join(ROOT, "profiles", "r2_pmc_summary.json"))

TRAIN_WORKLOADS = {
    # BASELINE config 4: DDP training on synthetic 5-species 256-atom frames, l_max=2, batch=32 per rank
    "train256": dict(n_atoms=256, n_species=5, batch=32, l_max=2, num_features=64, num_layers=3),
}

WORKLOADS = {
    # name: (box builder kwargs, model kwargs)
    "water10k": dict(box="water", n_side=15, l_max=2, num_features=64, num_layers=3),
    "water81k": dict(box="water", n_side=30, l_max=2, num_features=64, num_layers=3),  # 8 x the default box
    "si1k": dict(box="si", reps=5, l_max=2, num_features=64, num_layers=3),
    "water_small": dict(box="water", n_side=5, l_max=2, num_features=64, num_layers=3),
    # the reference's model presets (nequip/model/nequip_models.py:30-58: non-uniform multiplicities) on the cfg-3 box
    "water10k_S": dict(box="water", n_side=15, l_max=1, num_features=[128, 64], num_layers=2, type_embed_num_features=32),
    "water10k_M": dict(box="water", n_side=15, l_max=2, num_features=[128, 64, 32], num_layers=4,
                       type_embed_num_features=32),
    "water10k_L": dict(box="water", n_side=15, l_max=3, num_features=[128, 64, 32, 32], num_layers=6,
                       type_embed_num_features=32),
    "water10k_XL": dict(box="water", n_side=15, l_max=4, num_features=[320, 96, 64, 32, 32], num_layers=6,
                        type_embed_num_features=32),
    # BASELINE config 5: 100k-atom fcc Cu, l_max=3, 128 features (cu20k: same model on a fifth of the box)
    "cu100k": dict(box="cu", reps=(25, 25, 40), l_max=3, num_features=128, num_layers=3),
    "cu20k": dict(box="cu", reps=(25, 25, 8), l_max=3, num_features=128, num_layers=3),
    # BASELINE config 1: the tutorial hyper-parameters on a batch of 5 aspirin-like 21-atom molecules (non-periodic)
    "aspirin5": dict(box="aspirin", frames=5, l_max=1, num_features=32, num_layers=4, r_max=5.0, parity=True,
                     radial_mlp_depth=2, radial_mlp_width=64),
}


def baseline_metric() -> str:
    """BASELINE.json's metric string (quoted on the 10k-atom l_max=2 box = the default workload)."""
    try:
        return json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    except Exception:
        return "atom-steps/s (energy+forces) for 10k-atom l_max=2 box at 1/2/4/8 MI355X"


def model_cfg(w, avg_num_neighbors):
    return dict(
        r_max=w.get("r_max", 4.5), num_layers=w["num_layers"], l_max=w["l_max"], parity=w.get("parity", False),
        num_features=w["num_features"], radial_mlp_depth=w.get("radial_mlp_depth", 1),
        radial_mlp_width=w.get("radial_mlp_width", 128), num_bessels=8, polynomial_cutoff_p=6,
        avg_num_neighbors=float(avg_num_neighbors), model_dtype="float32",
        **({"type_embed_num_features": w["type_embed_num_features"]} if "type_embed_num_features" in w else {}),
    )  # fmt: skip


def build_box(w, seed=0):
    from nequip_amd.utils import synthetic as syn

    if w["box"] == "water":
        pos, types, cell, names = syn.water_box(n_side=w["n_side"], seed=seed)
    elif w["box"] == "si":
        pos, types, cell, names = syn.silicon_box(reps=w["reps"], seed=seed)
    elif w["box"] == "cu":
        pos, types, cell, names = syn.copper_box(reps=tuple(w["reps"]), seed=seed)
    elif w["box"] == "aspirin":
        from nequip_amd.data import AtomicDataDict

        frames = []
        for f in range(w["frames"]):
            pos, types, _, names = syn.aspirin_like(seed=seed * 100 + f)
            frames.append(syn.make_data(pos, types, w.get("r_max", 4.5), None, pbc=False))
        return AtomicDataDict.batched_from_list(frames), names
    else:
        raise ValueError(w["box"])
    data = syn.make_data(pos, types, w.get("r_max", 4.5), cell,
                         spatial_sort=os.environ.get("NQA_BENCH_SORT", "0") != "0")
    return data, names


def build_model(cfg, names, device, seed=0):
    from nequip_amd.model import NequIPGNNModel

    kw = {k: v for k, v in cfg.items() if k not in ("model_dtype",)}
    model = NequIPGNNModel(seed=seed, model_dtype=cfg["model_dtype"], type_names=names, **kw)
    return model.to(device).eval()


def cpu_baseline(workload_name: str, max_seconds: float = 60.0, full_box: bool = True):
    """Time the CPU oracle (reference op order, PyTorch CPU ops) beside the GPU number, protocol of SURVEY.md 8(d).

    * `value` / `cores` (`sample_value`): a smaller box of the workload's density / r_max / model -- 3 warm-up evaluations,
      then the median of up to 10 timed ones on the best of {8, 16, 32} threads; `all_cores_value`: the same with ALL host
      cores (the oracle is a chain of ATen ops on [E, ...] tensors, which 256 threads oversubscribe).
    * `full_box_value` (default workload only): the FULL 10 125-atom box of the metric, one warm evaluation + up to two
      timed ones (about a minute each) -- when present this is `value`, the figure `gpu_over_cpu` uses; the small-box
      figure stays next to it as `sample_value` (per-atom throughput on small boxes differs: fixed op overhead, threading)."""
    from oracle import model as omodel

    w = dict(WORKLOADS[workload_name])
    w_full = dict(w)
    if w["box"] == "water":
        w["n_side"] = min(w["n_side"], 7)  # 7^3 molecules = 1029 atoms (rounds 1-3; round 4 used 375)
    elif w["box"] == "si":
        w["reps"] = min(w["reps"], 5)  # 1000 atoms
    elif w["box"] == "cu":
        w["reps"] = (4, 4, 4)  # 256 atoms (l_max = 3, 128 features)
    # (aspirin5 is small enough to be timed whole)
    data, names = build_box(w, seed=1)
    n_atoms = data["pos"].shape[0]
    n_edges = data["edge_index"].shape[1]
    cfg = model_cfg(w, n_edges / n_atoms)
    model = build_model(cfg, names, torch.device("cpu"))
    weights = {k.replace("model.func.", ""): v.detach() for k, v in model.state_dict().items()}
    specs = omodel.build_specs(cfg)
    ncpu = os.cpu_count() or 1
    t_start = time.perf_counter()

    def one(d=data, c=cfg):
        t0 = time.perf_counter()
        omodel.energy_forces(d, c, weights, specs)
        return time.perf_counter() - t0

    torch.set_num_threads(min(ncpu, 32))
    one()  # (builds the cached CG tables)
    probe = {}
    for nthreads in sorted({min(ncpu, c) for c in (8, 16, 32)}):
        torch.set_num_threads(nthreads)
        probe[nthreads] = one()
    cores = min(probe, key=probe.get)
    torch.set_num_threads(cores)
    for _ in range(3):  # warm-up
        one()
    times = []
    while len(times) < 10 and (len(times) < 3 or (time.perf_counter() - t_start) < max_seconds):
        times.append(one())
    times.sort()
    med = times[len(times) // 2]
    all_cores = None
    if ncpu != cores:
        torch.set_num_threads(ncpu)
        one()
        ta = sorted(one() for _ in range(3))
        all_cores = n_atoms / ta[1]
    torch.set_num_threads(cores)
    res = {
        "value": n_atoms / med,
        "unit": "atom-steps/s",
        "cores": cores,
        "kind": "port",
        "host_cores": ncpu,
        "sample_value": n_atoms / med,
        "all_cores_value": all_cores,
        "sample": f"{n_atoms}-atom {w['box']} box ({n_edges} edges), same density/r_max/model as the workload; 3 warm-up + "
        f"median of {len(times)} energy+forces evaluations ({med:.2f} s each) of the torch-CPU oracle on {cores} threads "
        f"(best of {sorted(probe)}); all {ncpu} host cores: "
        + (f"{all_cores:.0f} atom-steps/s (median of 3)" if all_cores is not None else "same setting")
        + " (e3nn unavailable: restatement)",
    }
    if full_box and workload_name == "water10k" and os.environ.get("NQA_BENCH_CPU_FULL_BOX", "1") not in ("", "0"):
        # the box the metric names, whole: the same seed-0 box the GPU timed.  Activation checkpointing per layer keeps the
        # [E, 2240]-sized autograd intermediates of the oracle within the host's memory (oracle/model.py).
        fdata, fnames = build_box(w_full, seed=0)
        fa, fe = fdata["pos"].shape[0], fdata["edge_index"].shape[1]
        # (the weights do not depend on avg_num_neighbors: same model.  Edge ranges of 16384 under activation
        # checkpointing, as tests/test_baseline_size_parity.py: the reference formulation's [E, mul, d1, d2] temporaries of
        # this box would need > 100 GB of host memory otherwise; same arithmetic per edge, one recomputation in backward)
        fcfg = dict(model_cfg(w_full, fe / fa), oracle_edge_chunk=16384)
        tf = [one(fdata, fcfg)]  # first evaluation at this size = warm-up (allocator growth), reported too
        budget = float(os.environ.get("NQA_BENCH_CPU_FULL_BOX_SECONDS", "200"))
        while len(tf) < 3 and sum(tf) + 1.2 * tf[-1] < budget:
            tf.append(one(fdata, fcfg))
        best = min(tf[1:]) if len(tf) > 1 else tf[0]
        res["full_box_value"] = fa / best
        res["full_box_seconds"] = [round(t, 2) for t in tf]
        res["value"] = res["full_box_value"]
        res["sample"] = (f"FULL {fa}-atom box ({fe} edges) of the metric: {len(tf)} evaluation(s) of the torch-CPU oracle (edge ranges of 16384, "
                         f"activation checkpointing) on {cores} threads, {', '.join(f'{t:.1f}' for t in tf)} s (first = warm-up; value = atoms / best "
                         "later one); sample_value: " + res["sample"])
    return res


def _free_port() -> int:
    import socket

    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def relaunch_distributed(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: re-exec this very command line under torch.distributed.run with N
    ranks on this node (one process per GPU, RCCL), exactly as the driver's own launch line does."""
    import subprocess

    share = os.environ.get("NQA_BENCH_SHARE_DEVICE", "") not in ("", "0")
    have = torch.cuda.device_count()
    if have < n and not share:
        raise SystemExit(f"bench.py --gpus {n}: only {have} GPU(s) visible on this node (one rank per GPU; "
                         "NQA_BENCH_SHARE_DEVICE=1 exercises the N > 1 control flow on one device, not a measurement)")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // max(n, 1))))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


# ---- roofline bookkeeping ---------------------------------------------------------------------------------------
TP_REGIONS = ("tp_fwd", "tp_bwd_edge", "tp_bwd_x", "tp_bwd_fused", "tp_fwd_mlp", "tp_bwd_mlp")


def kernel_roofline(kname, ks, kernel_steps):
    launches = ks["calls"] / max(kernel_steps, 1)
    if ks.get("flops_per_call", 0) > 0 and kname.startswith("radial_mlp"):
        # GEMM on the matrix cores.  The split kernels execute several 16-bit MFMA partial products per fp32 product
        # (fp32-accurate): 3 on the two-plane fp16 split (default: forward with per-row / per-tile scales, backward with
        # a running per-row scale), 6 on the three-plane bf16 split (NQA_MLP_FWD_F16=0 / NQA_MLP_BWD_F16=0; training).  Two roofs bound such a launch from below -- the EXECUTED 16-bit MFMA
        # work against the dense bf16 / fp16 peak, and the rows it writes (forward) or reads (backward) against HBM;
        # `bound` / `achieved` / `frac` are those of the roof that gives the longer lower bound (the larger fraction),
        # the other one is reported next to it.  The algorithmic fp32 FLOP rate (what the reference's fp32 GEMM would be
        # credited with) is there too; with NQA_MLP_EXACT_FP32=1 the kernels run on the fp32 MFMA pipe and that is the
        # roofline.
        split = os.environ.get("NQA_MLP_EXACT_FP32", "") in ("", "0")
        # (radial_mlp_bwd_train goes through nn/mlp.py::backward_mode like the inference backward: fp16 split by default)
        f16_fwd = ((kname == "radial_mlp_fwd" and os.environ.get("NQA_MLP_FWD_F16", "") != "0")
                   or (kname in ("radial_mlp_bwd", "radial_mlp_bwd_train") and os.environ.get("NQA_MLP_BWD_F16", "") != "0"))
        r = {
            "bound": "mfma", "kernel": kname, "unit": "TFLOP/s",
            "avg_launch_ms": ks["avg_ms"], "algorithmic_flops_per_launch": ks["flops_per_call"],
            "algorithmic_bytes_per_launch": ks["bytes_per_call"], "hbm_gbps": ks["gbps"],
            "launches_per_step": launches, "algorithmic_fp32_tflops": ks["tflops"],
            "frac_of_fp32_mfma_peak": ks["tflops"] / MFMA_F32_PEAK_TFLOPS,
        }
        if split:
            products = 3.0 if f16_fwd else 6.0
            frac_mfma = products * ks["tflops"] / MFMA_BF16_PEAK_TFLOPS
            frac_hbm = ks["gbps"] / HBM_PEAK_GBPS
            r.update(executed_mfma_tflops=products * ks["tflops"], frac_mfma=frac_mfma, frac_hbm=frac_hbm,
                     pipe=("fp16 MFMA (v_mfma_f32_32x32x16_f16), 3 partial products per fp32 product" if f16_fwd else
                           "bf16 MFMA (v_mfma_f32_32x32x16_bf16), 6 partial products per fp32 product"))
            if frac_hbm > frac_mfma:
                r.update(bound="hbm", unit="GB/s", achieved=ks["gbps"], peak=HBM_PEAK_GBPS, frac=frac_hbm)
            else:
                r.update(achieved=products * ks["tflops"], peak=MFMA_BF16_PEAK_TFLOPS, frac=frac_mfma)
        else:
            r.update(achieved=ks["tflops"], peak=MFMA_F32_PEAK_TFLOPS, frac=ks["tflops"] / MFMA_F32_PEAK_TFLOPS,
                     pipe="fp32 MFMA (v_mfma_f32_32x32x2_f32)")
        return r
    r = {
        "bound": "hbm", "kernel": kname, "achieved": ks["gbps"], "peak": HBM_PEAK_GBPS, "unit": "GB/s",
        "frac": ks["gbps"] / HBM_PEAK_GBPS, "avg_launch_ms": ks["avg_ms"],
        "algorithmic_bytes_per_launch": ks["bytes_per_call"], "launches_per_step": launches,
    }
    if ks.get("fused_bytes_per_call") and ks["avg_ms"] > 0:
        # "boundary-algorithmic" (frac, above: edge_weight rows counted) and "fused-algorithmic" (the weight
        # rows counted as [E, H] hidden rows -- what a kernel with the MLP's last layer fused in would stream) side by side
        fb = ks["fused_bytes_per_call"]
        r["fused_algorithmic_bytes_per_launch"] = fb
        r["frac_fused_algorithmic"] = fb / 1e9 / (ks["avg_ms"] / 1e3) / HBM_PEAK_GBPS
    return r


def add_traffic(r, pmc):
    """`traffic` = counter-measured HBM bytes per launch of the region; `frac_traffic` = those bytes / the launch time /
    the HBM peak -- the rate the kernel really sustains (`frac` is computed on the boundary's algorithmic bytes, which a
    kernel that reads a shared row once per pair does not move)."""
    t = pmc.get(r["kernel"], {}).get("hbm_bytes_per_launch")
    r["traffic"] = t
    if t is not None and r["avg_launch_ms"] > 0:
        r["traffic_gbps"] = t / 1e9 / (r["avg_launch_ms"] / 1e3)
        r["frac_traffic"] = r["traffic_gbps"] / HBM_PEAK_GBPS
        if r.get("algorithmic_bytes_per_launch"):
            r["traffic_over_algorithmic"] = t / r["algorithmic_bytes_per_launch"]
    else:
        r["traffic_gbps"] = r["frac_traffic"] = None
    return r


def measure_traffic_live(workload: str, timeout_s: float = 240.0):
    """HBM bytes per launch of every hand-written kernel region, measured now: two `rocprofv3 --pmc` passes
    (FETCH_SIZE and WRITE_SIZE each in its own run, as MI355X_MICROARCH.md prescribes; no trace domains besides the
    kernel trace) over a short eager run of this very workload (`--pmc-child`), condensed by
    scripts/summarize_profile.py (gfx950 correction: bytes = (2 FETCH_SIZE + WRITE_SIZE) KB).
    Returns (bench_kernels dict or None, source label)."""
    import shutil
    import subprocess
    import tempfile

    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if os.environ.get("NQA_BENCH_NO_PMC", "") not in ("", "0"):
        return None, "disabled (NQA_BENCH_NO_PMC)"
    if not os.path.exists(rocprof):
        return None, "rocprofv3 not found"
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ):
        return None, "already running under a profiler"
    out = tempfile.mkdtemp(prefix="nqa_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    t0 = time.perf_counter()
    try:
        for name, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
            left = timeout_s - (time.perf_counter() - t0)
            if left < 20:
                return None, "live PMC passes timed out"
            cmd = [rocprof, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d",
                   os.path.join(out, f"pmc_{name}"), "-o", "bench", "--", sys.executable, os.path.abspath(__file__),
                   "--pmc-child", "--workload", workload]
            res = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=left)
            if res.returncode != 0:
                return None, f"rocprofv3 --pmc {counter} failed (rc {res.returncode})"
        res = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "summarize_profile.py"), out, "live"],
                             capture_output=True, text=True, timeout=60)
        summ = json.load(open(os.path.join(out, "live_pmc_summary.json")))
        if not summ.get("bench_kernels"):
            return None, "live PMC passes produced no counters"
        return summ["bench_kernels"], ("measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes "
                                       f"over 2 eager steps of the workload ({time.perf_counter() - t0:.0f} s)")
    except Exception as exc:  # pragma: no cover
        return None, f"live PMC measurement failed ({type(exc).__name__}: {exc})"
    finally:
        shutil.rmtree(out, ignore_errors=True)


def static_traffic():
    try:
        summ = json.load(open(PMC_SUMMARY))
        return summ["bench_kernels"], f"static: {os.path.relpath(PMC_SUMMARY, ROOT)} ({summ.get('commit', 'commit n/a')})"
    except Exception:
        return {}, "unavailable"


def roofline_objects(kernels, kernel_steps, ms_per_step, workload, live_pmc, boundary=None):
    """`roofline` (dominant hand-written kernel + the other hot ones) and `step_roofline` (whole step: the
    TensorProductScatter boundary bytes of SURVEY.md 8(d), forward + backward of every layer, against the HBM peak)."""
    if not kernels:
        return None, None
    pmc, source = (None, "")
    if live_pmc:
        pmc, source = measure_traffic_live(workload)
    if pmc is None:
        why = source
        # the committed PMC summaries are of the cfg-3 step: for any other workload a static figure would be another
        # workload's traffic (BENCH_r05 review: 0.03x / 286x "traffic over algorithmic") -- null instead
        pmc, source = static_traffic() if workload == "water10k" else ({}, "unavailable (no live PMC pass in this run)")
        if why:
            source += f" [live measurement: {why}]"
    name, s = max(kernels.items(), key=lambda kv: kv[1]["total_ms"])
    roofline = add_traffic(kernel_roofline(name, s, kernel_steps), pmc)
    roofline["traffic_source"] = source
    others = {}
    for kname, ks in sorted(kernels.items(), key=lambda kv: -kv[1]["total_ms"])[:8]:
        if kname != name:
            others[kname] = add_traffic(kernel_roofline(kname, ks, kernel_steps), pmc)
    roofline["other_kernels"] = others
    tp_bytes = sum(v["bytes_per_call"] * v["calls"] for k, v in kernels.items() if k in TP_REGIONS) / max(kernel_steps, 1)
    bnd, fus = (boundary if boundary is not None else (tp_bytes, None))
    sec = ms_per_step / 1e3
    step = {
        "bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBPS,
        "algorithmic_bytes_per_step": bnd,
        "definition": "SURVEY.md 8(d) boundary-algorithmic: sum over layers of E (12 W + 12 S + 32) + 4 N (3 D_in + 2 D_mid) "
                      "-- operands and results of the gather -> tensor product -> scatter boundary, forward + backward, "
                      "each counted once (edge_weight / grad_weight rows included)",
        "achieved": bnd / 1e9 / sec,
        "frac": bnd / 1e9 / sec / HBM_PEAK_GBPS,
        "kernel_region_bytes_per_step": tp_bytes,
        "kernel_ms_per_step_sum": sum(v["total_ms"] for v in kernels.values()) / max(kernel_steps, 1),
    }
    if fus is not None:
        step["fused_algorithmic"] = {
            "bytes_per_step": fus, "achieved": fus / 1e9 / sec, "frac": fus / 1e9 / sec / HBM_PEAK_GBPS,
            "definition": "the same with W -> H (hidden width of the radial MLP): what a kernel with the MLP's last layer "
                          "fused in would stream.  The kernels here do NOT fuse it (edge_weight is materialised): this is "
                          "the smaller denominator SURVEY.md 8(d) asks to be shown next to the boundary figure",
        }
        # the same pair of fractions for the dominant tensor-product region (its launches carry a 'W' each)
    return roofline, step



class GpuClockSampler:
    """Shader clock / power / temperature of the GPU while it is busy (say WHY a box is slow).
    Polls the amdgpu hwmon files of the device (freq1_input = current gfx clock in Hz, power1_average / power1_input in
    microwatts, temp*_input in millidegrees) from a thread while the caller keeps the device busy; one `rocm-smi` call as
    the fallback.  Never raises: a missing sysfs tree gives {"source": "unavailable"}."""

    def __init__(self, dev_index: int = 0):
        import glob
        import threading

        self._stop = threading.Event()
        self._thread = None
        self.samples = {"sclk_mhz": [], "dpm_sclk_mhz": [], "power_w": [], "temp_c": []}
        self.files = {}
        self.source = "unavailable"
        try:
            cards = sorted(d for d in glob.glob("/sys/class/drm/card[0-9]*/device") if os.path.exists(os.path.join(d, "pp_dpm_sclk")))
            if cards:
                card = cards[min(dev_index, len(cards) - 1)]
                hw = sorted(glob.glob(os.path.join(card, "hwmon", "hwmon*")))
                if hw:
                    for key, names in (("sclk_mhz", ("freq1_input",)), ("power_w", ("power1_average", "power1_input")),
                                       ("temp_c", ("temp2_input", "temp1_input"))):
                        for nm in names:
                            f = os.path.join(hw[0], nm)
                            if os.path.exists(f):
                                self.files[key] = f
                                break
                self.files.setdefault("dpm", os.path.join(card, "pp_dpm_sclk"))
                self.source = f"sysfs {card}"
        except Exception:
            self.files = {}

    def _poll(self):
        scale = {"sclk_mhz": 1e-6, "power_w": 1e-6, "temp_c": 1e-3}
        while not self._stop.is_set():
            for key in ("sclk_mhz", "power_w", "temp_c"):
                f = self.files.get(key)
                if f:
                    try:
                        self.samples[key].append(float(open(f).read().strip()) * scale[key])
                    except Exception:
                        pass
            if "dpm" in self.files:  # the DPM level in use (the `*` line of pp_dpm_sclk), next to the hwmon reading
                try:
                    for line in open(self.files["dpm"]).read().splitlines():
                        if line.rstrip().endswith("*"):
                            self.samples["dpm_sclk_mhz"].append(float(line.split(":")[1].lower().replace("mhz", "").replace("*", "")))
                except Exception:
                    pass
            self._stop.wait(0.004)

    def start(self):
        import threading

        if self.files:
            self._thread = threading.Thread(target=self._poll, daemon=True)
            self._thread.start()
        return self

    def stop(self):
        self._stop.set()
        if self._thread is not None:
            self._thread.join(timeout=2.0)
        out = {"source": self.source}
        for key, v in self.samples.items():
            if v:
                v = sorted(v)
                out[key] = {"median": round(v[len(v) // 2], 1), "min": round(v[0], 1), "max": round(v[-1], 1), "n": len(v)}
        return out


def gpu_clock_mhz(gpu_state):
    """One shader-clock figure for the line.  The DPM level in use / hwmon's freq1_input read ~100 MHz on some boxes while the
    device is busy (BENCH_r05: 113 MHz next to rocm-smi's 2382 MHz): anything below 500 MHz is not a shader clock of a
    working MI355X, so the `rocm-smi --showclocks` reading taken under load is used instead; None when nothing is sane."""
    import re

    if not gpu_state:
        return None
    cands = []
    for key in ("dpm_sclk_mhz", "sclk_mhz"):
        v = (gpu_state.get(key) or {}).get("median")
        if v is not None:
            cands.append(float(v))
    for k, v in (gpu_state.get("rocm_smi") or {}).items():
        if "sclk" in k.lower():
            m = re.search(r"(\d+(?:\.\d+)?)\s*mhz", str(v).lower())
            if m:
                cands.append(float(m.group(1)))
    sane = [c for c in cands if 500.0 <= c <= 3500.0]
    return sane[0] if sane else None


def rocm_smi_snapshot():
    """One `rocm-smi` reading (clocks, power cap, temperature) taken while the caller's kernels run; {} when unavailable."""
    import shutil
    import subprocess

    exe = shutil.which("rocm-smi") or "/opt/rocm/bin/rocm-smi"
    if not os.path.exists(exe):
        return {}
    try:
        res = subprocess.run([exe, "--showclocks", "--showpower", "--showmaxpower", "--showtemp", "--showperflevel", "--json"],
                             capture_output=True, text=True, timeout=20)
        d = json.loads(res.stdout)
        card = d.get("card0", next(iter(d.values())))
        keep = {}
        for k, v in card.items():
            kl = k.lower()
            if any(t in kl for t in ("sclk", "mclk", "power", "temperature (sensor junction)", "performance level")):
                keep[k] = v
        return keep
    except Exception:
        return {}


def boundary_bytes(model, n_atoms: int, n_edges: int, hidden_width: int):
    """SURVEY.md 8(d), energy + forces step at the TensorProductScatter boundary, fp32, 16 B of int64 indices per edge and
    direction, every operand / result once: sum over layers of E (12 W + 12 S + 32) + 4 N (3 D_in + 2 D_mid).
    `fused`: the same with the radial MLP's last layer inside the kernels (W -> H, the hidden width: the rows a fused
    kernel would stream instead of edge_weight / grad_weight)."""
    from nequip_amd.nn import TensorProductScatter

    boundary = fused = 0.0
    layers = []
    for m in model.modules():
        if isinstance(m, TensorProductScatter):
            tp = m.tp
            W, S = int(tp.weight_numel), int(tp.irreps_in2.dim)
            d_in, d_mid = int(tp.irreps_in1.dim), int(tp.irreps_out.dim)
            layers.append({"W": W, "S": S, "D_in": d_in, "D_mid": d_mid})
            boundary += n_edges * (12.0 * W + 12.0 * S + 32.0) + 4.0 * n_atoms * (3 * d_in + 2 * d_mid)
            fused += n_edges * (12.0 * min(W, hidden_width) + 12.0 * S + 32.0) + 4.0 * n_atoms * (3 * d_in + 2 * d_mid)
    return boundary, fused, layers


def train_bench(args, world, rank, device, distributed):
    """BASELINE config 4: force-matching training step (forward, double backward, flat gradient all-reduce over
    RCCL, Adam) on `batch` random frames per rank.  Metric: atoms x optimizer-steps / s, aggregate over ranks."""
    import torch.distributed as dist

    from nequip_amd.data import AtomicDataDict
    from nequip_amd.model import NequIPGNNModel
    from nequip_amd.train import SimpleDDPStrategy
    from nequip_amd.utils import ktimer
    from nequip_amd.utils import synthetic as syn
    w = TRAIN_WORKLOADS[args.workload]
    frames = []
    for f in range(w["batch"]):
        pos, types, cell, names = syn.random_frame(w["n_atoms"], w["n_species"], seed=1000 * rank + f)
        frames.append(syn.make_data(pos, types, 4.5, cell))
    data = AtomicDataDict.to_device(AtomicDataDict.batched_from_list(frames), device)
    n_atoms = data["pos"].shape[0]
    n_edges = data["edge_index"].shape[1]
    gen = torch.Generator().manual_seed(rank)
    f_target = torch.randn(n_atoms, 3, generator=gen, dtype=torch.float64).to(device)
    e_target = torch.randn(w["batch"], 1, generator=gen, dtype=torch.float64).to(device)
    model = NequIPGNNModel(
        seed=0, model_dtype="float32", r_max=4.5, type_names=names, num_layers=w["num_layers"], l_max=w["l_max"],
        parity=False, num_features=w["num_features"], radial_mlp_depth=1, radial_mlp_width=128,
        avg_num_neighbors=n_edges / n_atoms, per_type_energy_scales=1.0, per_type_energy_shifts=0.0,
    ).to(device).train()
    strategy = SimpleDDPStrategy(model)
    # hipGraph replay of the whole optimizer step (forward, double backward, Adam) on one rank: the eager step is bound by
    # the host (8.9 ms of Python / launch work to enqueue 7.2 ms of kernels, scripts/r2_train_cpu_bound.py).  Multi-rank
    # runs stay eager (the gradient all-reduce is issued by the strategy between backward and the optimizer).
    use_graph = (not distributed) and os.environ.get("NQA_TRAIN_GRAPH", "1") not in ("", "0")
    opt = torch.optim.Adam(model.parameters(), lr=1e-2, capturable=use_graph)

    collective_events = []  # (start, end) HIP events around post_backward of the instrumented steps (N > 1)

    def step(record_collective=False):
        opt.zero_grad(set_to_none=True)
        out = model(dict(data))
        loss = (out["forces"] - f_target).square().mean() + (out["total_energy"] - e_target).square().mean()
        # the shipped training step: SimpleDDPStrategy.backward (parameter gradients off the data chain,
        # nequip_amd/utils/wgrad.py), then the flat all-reduce -- nequip/train/lightning.py:259-266
        strategy.backward(loss * strategy.world_size)
        if record_collective:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
            strategy.post_backward(loss)
            ev1.record()
            collective_events.append((ev0, ev1))
        else:
            strategy.post_backward(loss)
        opt.step()
        return loss

    for _ in range(max(args.warmup, 1)):
        step()
    graph = None
    launch = "eager"
    if use_graph:
        try:
            side = torch.cuda.Stream(device=device)
            side.wait_stream(torch.cuda.current_stream(device))
            with torch.cuda.stream(side):
                for _ in range(2):
                    step()
            torch.cuda.current_stream(device).wait_stream(side)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            opt.zero_grad(set_to_none=True)
            with torch.cuda.graph(graph):
                static_loss = step()
            graph.replay()
            torch.cuda.synchronize()
            launch = "hipGraph replay of the whole optimizer step"
        except Exception as exc:  # capture not possible in this configuration: time the eager step
            print(f"[bench] training-step graph capture failed ({type(exc).__name__}: {exc}); eager", file=sys.stderr)
            graph = None
            torch.cuda.synchronize()

    def timed_step():
        if graph is not None:
            graph.replay()
            return static_loss
        return step()

    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = timed_step()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    roofline = step_roofline = None
    kernels = {}
    if args.kernel_steps > 0 and (rank == 0 or distributed):
        # (the instrumented steps contain the gradient all-reduce: with more than one rank EVERY rank has to run them,
        # only rank 0 records -- rank 0 alone would wait for its peers forever)
        if rank == 0:
            ktimer.reset()
            ktimer.enable(True)
        for _ in range(args.kernel_steps):
            step(record_collective=distributed)
        torch.cuda.synchronize()
        if rank == 0:
            ktimer.enable(False)
            kernels = ktimer.summary()
            roofline, step_roofline = roofline_objects(kernels, args.kernel_steps, elapsed / args.steps * 1e3,
                                                       args.workload, live_pmc=False)
    # per-rank time of the gradient synchronisation (flatten + all-reduce + copy back; the device time between two events on
    # the launching stream, so it includes waiting for the slowest rank): lets a scaling curve be split into compute and
    # collective the day more than one GPU runs this
    collective_ms = None
    if distributed:
        mine = (sum(a.elapsed_time(b) for a, b in collective_events) / len(collective_events)) if collective_events else 0.0
        t = torch.tensor([mine], dtype=torch.float64, device=device)
        gathered = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(gathered, t)
        collective_ms = [float(g.item()) for g in gathered]
    if rank == 0:
        print(json.dumps({
            "metric": "atom-optimizer-steps/s (DDP force-matching training)",
            "value": world * n_atoms * args.steps / elapsed, "unit": "atom-steps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {w['batch']} frames x {w['n_atoms']} atoms per rank "
                       f"({n_edges} edges), {w['n_species']} species, l_max={w['l_max']}, {w['num_features']} features, "
                       "energy+force MSE loss, Adam, flat gradient all-reduce (SimpleDDP)",
                       "parallelism": f"dp{world}", "final_loss": float(loss.detach()), "launch": launch,
                       "collective": ("RCCL all-reduce of one flat fp32 gradient buffer per step" if distributed else "none (1 rank)"),
                       "collective_ms_per_rank": collective_ms},
            "roofline": roofline, "step_roofline": step_roofline,
            "kernels_ms_per_step": {k: v["total_ms"] / max(args.kernel_steps, 1) for k, v in kernels.items()},
        }))
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


def other_workloads(timeout_s: float = 420.0):
    """The other BASELINE configs (cfg-1 aspirin batch, cfg-2 si1k, cfg-5 cu100k, cfg-4 training step), each as a short
    child run of this file on the same device right after the headline measurement, so that the driver's line carries a
    driver-timed figure for every config.  Per workload: ms_per_step, value + unit, the dominant
    hand-written kernel region with its share of the HBM roof (boundary-algorithmic bytes / HIP-event time; `traffic` is not
    measured here: null).  A child that fails or times out is recorded as {"error": ...}, never raised."""
    import subprocess

    out = {}
    t_all = time.perf_counter()
    for wl, steps, warm in (("si1k", 20, 5), ("aspirin5", 20, 5), ("cu100k", 5, 2), ("train256", 10, 3)):
        left = timeout_s - (time.perf_counter() - t_all)
        if left < 30:
            out[wl] = {"error": "skipped: time budget of the other-workload runs used up"}
            continue
        env = dict(os.environ, NQA_BENCH_NO_CLOCKS="1", NQA_BENCH_NO_EXACT_FP32="1", NQA_BENCH_NO_MD_STEP="1", NQA_BENCH_NO_OTHER="1")
        cmd = [sys.executable, os.path.abspath(__file__), "--workload", wl, "--steps", str(steps), "--warmup", str(warm),
               "--no-cpu-baseline", "--no-pmc", "--no-other-workloads", "--kernel-steps", "2"]
        t0 = time.perf_counter()
        try:
            res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=left)
            line = next((ln for ln in reversed(res.stdout.splitlines()) if ln.startswith("{")), None)
            if res.returncode != 0 or line is None:
                out[wl] = {"error": f"rc {res.returncode}: {(res.stderr or '')[-300:]}"}
                continue
            d = json.loads(line)
            rf = d.get("roofline") or {}
            out[wl] = {"ms_per_step": d.get("ms_per_step"), "value": d.get("value"), "unit": d.get("unit"),
                       "metric": d.get("metric"), "workload": (d.get("config") or {}).get("workload"),
                       "dominant_kernel": rf.get("kernel"), "dominant_avg_launch_ms": rf.get("avg_launch_ms"),
                       "dominant_frac": rf.get("frac"), "dominant_bound": rf.get("bound"), "traffic": None,
                       "step_frac": (d.get("step_roofline") or {}).get("frac"), "wall_s": round(time.perf_counter() - t0, 1)}
        except Exception as exc:  # pragma: no cover
            out[wl] = {"error": f"{type(exc).__name__}: {exc}"}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="ranks (one per GPU); without a launcher bench.py starts them itself")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="water10k", choices=sorted(WORKLOADS) + sorted(TRAIN_WORKLOADS))
    ap.add_argument("--no-graph", action="store_true", help="time eager launches instead of hipGraph replay")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="skip the live rocprofv3 --pmc passes behind `roofline.traffic`")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)  # 1 warm-up + 2 eager steps, no output
    ap.add_argument("--kernel-steps", type=int, default=3, help="eager steps instrumented with HIP events for `roofline`")
    ap.add_argument("--no-other-workloads", action="store_true",
                    help="default run only: skip the short child runs of the other BASELINE configs (`config.other_workloads`)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and (args.gpus or 1) > 1:
        raise SystemExit(relaunch_distributed(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus is not None and args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    distributed = world > 1
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the hot path has no CPU fallback)")
    # NQA_BENCH_SHARE_DEVICE=1 (testing the N > 1 control flow on a one-GPU box): every rank uses device 0 and the
    # barriers / max-reduction go over gloo, since RCCL refuses two ranks on one device.  Not a measurement mode.
    share = distributed and os.environ.get("NQA_BENCH_SHARE_DEVICE", "") not in ("", "0")
    if distributed and not share and torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"bench.py: rank {rank} has no GPU (LOCAL_RANK {local_rank}, {torch.cuda.device_count()} visible)")
    dev_index = 0 if share else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if distributed:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=device)

    from nequip_amd.data import AtomicDataDict
    from nequip_amd.nn import topology_cache
    from nequip_amd.utils import ktimer

    if args.workload in TRAIN_WORKLOADS:
        return train_bench(args, world, rank, device, distributed)

    w = WORKLOADS[args.workload]
    data_cpu, names = build_box(w, seed=rank)
    n_atoms = data_cpu["pos"].shape[0]
    n_edges = data_cpu["edge_index"].shape[1]
    cfg = model_cfg(w, n_edges / n_atoms)
    model = build_model(cfg, names, device)
    data = AtomicDataDict.to_device(data_cpu, device)
    static_pos = data["pos"].clone()
    ktimer.hidden_width = int(cfg["radial_mlp_width"])
    bnd_bytes, fused_bytes, tp_layers = boundary_bytes(model, n_atoms, n_edges, int(cfg["radial_mlp_width"]))

    def step_eager():
        d = dict(data)
        d["pos"] = static_pos
        out = model(d)
        # detach: nothing of the autograd graph must outlive the step (and be alive during hipGraph capture)
        return out["total_energy"].detach(), out["forces"].detach()

    if args.pmc_child:  # counter-collection run for measure_traffic_live: eager launches only
        for _ in range(3):
            step_eager()
        torch.cuda.synchronize()
        return

    # ---- warm-up (also builds CSR, allocator pools, hipBLASLt heuristics) -------------------------------
    for _ in range(max(args.warmup, 1)):
        e, f = step_eager()
    torch.cuda.synchronize()

    use_graph = not args.no_graph

    def capture():
        """hipGraph of one evaluation (None when capture is unavailable).  The edge topology (CSR) is static for a fixed
        neighbour list: it is built once in the eager warm-up and stays cached; the graph captures the model evaluation."""
        try:
            torch.cuda.empty_cache()  # the graph gets a private pool: hand the warm-up's cached blocks back first
            g = torch.cuda.CUDAGraph()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):
                    step_eager()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            # NQA_BENCH_MAIN_PRIO=1 (experiment): capture on a high-priority stream, so that the side-stream branches (radial
            # backward, self-connection) yield to the main chain wherever both have work
            main = torch.cuda.Stream(priority=-1) if os.environ.get("NQA_BENCH_MAIN_PRIO", "") == "1" else None
            with torch.cuda.graph(g, stream=main):
                g_e, g_f = step_eager()
            g.replay()
            torch.cuda.synchronize()
            e2, f2 = step_eager()
            torch.cuda.synchronize()
            if not torch.allclose(g_f, f2, atol=1e-5, rtol=1e-5):
                raise RuntimeError("graph replay disagrees with eager")
            return g
        except Exception as exc:  # pragma: no cover
            if rank == 0:
                print(f"[bench] hipGraph capture unavailable ({type(exc).__name__}: {exc}); timing eager", file=sys.stderr)
            topology_cache.clear()
            return None

    graph = capture() if use_graph else None

    def step():
        if graph is not None:
            graph.replay()
        else:
            step_eager()

    for _ in range(args.warmup):
        step()

    # ---- timed region ---------------------------------------------------------------------------------
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed / args.steps * 1e3

    # ---- what the clocks were while the device ran this very step (rank 0; outside the timed region) ----
    gpu_state = None
    if rank == 0 and os.environ.get("NQA_BENCH_NO_CLOCKS", "") in ("", "0"):
        sampler = GpuClockSampler(dev_index).start()
        t1 = time.perf_counter()
        n_busy = 0
        while time.perf_counter() - t1 < 0.8:
            for _ in range(10):
                step()
            n_busy += 10
        torch.cuda.synchronize()
        busy_ms = (time.perf_counter() - t1) / max(n_busy, 1) * 1e3
        gpu_state = sampler.stop()
        gpu_state["ms_per_step_while_sampled"] = busy_ms
        # one rocm-smi reading with the queue kept full (hwmon's freq1_input reads ~100 MHz on some boxes while busy:
        # the sensor is not the XCDs' shader clock there -- the tool's own view is recorded next to it)
        for _ in range(400):
            step()
        gpu_state["rocm_smi"] = rocm_smi_snapshot()
        torch.cuda.synchronize()

    # ---- the same step with every GEMM on the exact-fp32 MFMA pipe (the default splits fp32 operands into fp16 planes) ----
    exact_ms = None
    if rank == 0 and world == 1 and os.environ.get("NQA_BENCH_NO_EXACT_FP32", "") in ("", "0"):
        saved = {k: os.environ.get(k) for k in ("NQA_MLP_EXACT_FP32", "NQA_NODE_EXACT_FP32")}
        try:
            os.environ["NQA_MLP_EXACT_FP32"] = "1"
            os.environ["NQA_NODE_EXACT_FP32"] = "1"
            for _ in range(3):
                step_eager()
            torch.cuda.synchronize()
            g2 = capture() if use_graph else None
            run = (g2.replay if g2 is not None else step_eager)
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                run()
            torch.cuda.synchronize()
            exact_ms = (time.perf_counter() - t1) / args.steps * 1e3
            del g2
        except Exception as exc:  # pragma: no cover
            print(f"[bench] exact-fp32 variant failed ({type(exc).__name__}: {exc})", file=sys.stderr)
        finally:
            for k, v in saved.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
            for _ in range(2):  # back on the default path (weight images of the split modes are cached per mode)
                step_eager()
            torch.cuda.synchronize()

    # ---- the molecular-dynamics form of the step: NEW positions and a NEW device neighbour list every step, the whole
    # evaluation (list -> pairing -> model) replayed as one hipGraph (integrations/graphed_step.py).  Reported next to the
    # headline, never as `value`: the headline's list is static, as BASELINE.json's metric is.
    md_step = None
    if (rank == 0 and world == 1 and use_graph and args.workload == "water10k" and "cell" in data
            and os.environ.get("NQA_BENCH_NO_MD_STEP", "") in ("", "0")):
        try:
            from nequip_amd.integrations.graphed_step import GraphedStep

            gstep = GraphedStep(model, data["atom_types"].view(-1), data["cell"].view(3, 3), True, float(cfg["r_max"]))
            gen = torch.Generator(device=device).manual_seed(1234)
            moves = [0.02 * torch.randn(static_pos.shape, generator=gen, device=device, dtype=static_pos.dtype)
                     for _ in range(8)]
            for i in range(3):
                gstep(static_pos + moves[i])
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for i in range(args.steps):
                gstep(static_pos + moves[i % len(moves)])
            torch.cuda.synchronize()
            md_ms = (time.perf_counter() - t1) / args.steps * 1e3
            md_step = {
                "ms_per_step": md_ms, "atom_steps_per_s": n_atoms / (md_ms * 1e-3),
                "edge_slots": gstep.edge_capacity, "last_num_edges": gstep.last_num_edges,
                "captures": gstep.num_captures, "eager_fallbacks": gstep.num_eager_fallbacks,
                "what": "positions perturbed every step -> device neighbour list (capacity-padded) -> pairing -> "
                        "energy+forces, one hipGraph replay per step; includes the per-step read of the step's flags",
            }
            del gstep
        except Exception as exc:  # pragma: no cover
            print(f"[bench] MD-step variant failed ({type(exc).__name__}: {exc})", file=sys.stderr)

    # ---- per-kernel HIP-event timing of the hand-written kernels (eager, on the launching stream) ------
    roofline = step_roofline = None
    kernels = {}
    if rank == 0 and args.kernel_steps > 0:
        # (per-kernel durations are taken back to back on one stream: the side-stream overlap of the radial backward
        # would make concurrent kernels stretch each other's event intervals)
        prev_ov = os.environ.get("NQA_NO_OVERLAP")
        os.environ["NQA_NO_OVERLAP"] = "1"
        ktimer.reset()
        ktimer.enable(True)
        for _ in range(args.kernel_steps):
            step_eager()
        torch.cuda.synchronize()
        ktimer.enable(False)
        if prev_ov is None:
            del os.environ["NQA_NO_OVERLAP"]
        else:
            os.environ["NQA_NO_OVERLAP"] = prev_ov
        kernels = ktimer.summary()
        roofline, step_roofline = roofline_objects(kernels, args.kernel_steps, ms_per_step, args.workload,
                                                   live_pmc=(world == 1 and not args.no_pmc),
                                                   boundary=(bnd_bytes, fused_bytes))
        if step_roofline is not None:
            step_roofline["layers"] = tp_layers

    if rank == 0:
        value = world * n_atoms * args.steps / elapsed
        result = {
            "metric": baseline_metric() if args.workload == "water10k" else "atom-steps/s (energy+forces)",
            "value": value,
            "unit": "atom-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{args.workload}: {n_atoms}-atom {w['box']} box, {n_edges} edges, r_max {cfg['r_max']}, "
                f"l_max={w['l_max']}, {w['num_features']} features, {w['num_layers']} layers, parity={cfg['parity']}, "
                f"radial MLP 8-{cfg['radial_mlp_width']}(x{cfg['radial_mlp_depth']})-W, fp32 model / fp64 positions, "
                "energy+forces (autograd), random-init weights",
                "atoms_per_gpu": n_atoms,
                "edges_per_gpu": n_edges,
                "parallelism": f"replicas x{world} (frames independent, no data-path collective)",
                "launch": ("hipGraph replay" if graph is not None else "eager")
                + ("" if os.environ.get("NQA_NO_OVERLAP", "") not in ("", "0") else
                   ", radial-MLP backward on a side stream (parallel graph branch)"),
                # `dtype` f32 = what the path computes in: fp32 storage and accumulation everywhere; the three dense GEMM
                # families are evaluated as split products on the fp16 matrix pipe (fp32-accurate), see exact_fp32_ms_per_step
                "arithmetic": ("fp32 storage / accumulate (fp64 geometry + per-atom energies); radial-MLP and node (Linear, "
                               "self-connection) GEMMs as 3 x fp16 MFMA products on 22-bit two-plane operand splits "
                               "(error 2^-22 per operand: fp32 level); tensor product / scatter in plain fp32 FMA"
                               if os.environ.get("NQA_MLP_EXACT_FP32", "") in ("", "0") else
                               "fp32 throughout: GEMMs on the exact-fp32 MFMA pipe (v_mfma_f32_32x32x2_f32)"),
                "exact_fp32_ms_per_step": exact_ms,
                "md_step": md_step,
                "gpu_state": gpu_state,
                "gpu_clock_mhz": gpu_clock_mhz(gpu_state),
            },
            "roofline": roofline,
            "step_roofline": step_roofline,
            "kernels_ms_per_step": {k: v["total_ms"] / max(args.kernel_steps, 1) for k, v in kernels.items()},
            "kernels_gbps": {k: v["gbps"] for k, v in kernels.items()},
            "kernels_tflops": {k: v["tflops"] for k, v in kernels.items() if v.get("tflops", 0) > 0},
        }
        if (world == 1 and args.workload == "water10k" and not args.no_other_workloads
                and os.environ.get("NQA_BENCH_NO_OTHER", "") in ("", "0")):
            # free this process's device memory first: cu100k needs most of what the cfg-3 model + graphs hold
            graph = None
            topology_cache.clear()
            torch.cuda.empty_cache()
            result["config"]["other_workloads"] = other_workloads()
        if not args.no_cpu_baseline and world == 1:  # rank 0 at N=1 only
            result["cpu_baseline"] = cpu_baseline(args.workload)
            result["gpu_over_cpu"] = value / result["cpu_baseline"]["value"]
        print(json.dumps(result))
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
