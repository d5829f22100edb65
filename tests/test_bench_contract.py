"""bench.py prints ONE JSON line with the fields the driver and the judge read (task contract): metric / value / unit /
n_gpus / steps / warmup / ms_per_step / higher_is_better / scaling / vs_baseline / dtype / data / config.workload, plus
`roofline` (bound, achieved, peak, unit, frac, traffic) and `cpu_baseline` (value, unit, cores, kind, sample)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


@pytest.mark.gpu
def test_bench_line_on_a_small_box(device):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "water_small", "--steps", "3",
                        "--warmup", "1", "--kernel-steps", "1", "--no-pmc", "--no-cpu-baseline"], env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line"
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "step_roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["unit"] == "atom-steps/s" and d["dtype"] == "f32" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - d["config"]["atoms_per_gpu"] / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_ms"):
        assert k in rf, k
    assert rf["bound"] in ("hbm", "mfma") and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    # every hand-written kernel family of the path ran (no region missing = nothing fell off the HIP path)
    ran = set(d["kernels_ms_per_step"])
    assert {"tp_fwd", "radial_mlp_fwd", "radial_mlp_bwd", "node_linear", "node_fused", "edge_embed_fwd",
            "edge_embed_bwd"} <= ran  # (node_fused: Gate + linear_1 + self-connection of a layer boundary, one launch)
    assert ran & {"tp_bwd_fused", "tp_bwd_edge"}


@pytest.mark.gpu
def test_bench_line_traffic_is_measured_for_the_dominant_kernel(device):
    """`roofline.traffic` must be populated from the live rocprofv3 --pmc passes for the dominant kernel and for every
    tensor-product region next to it (a stale kernel-name pattern had silently emptied it)."""
    import shutil

    if not (shutil.which("rocprofv3") or os.path.exists("/opt/rocm/bin/rocprofv3")):
        pytest.skip("rocprofv3 not installed")
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK") and not k.startswith(("ROCPROF", "ROCP_"))}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "water_small", "--steps", "3",
                        "--warmup", "1", "--kernel-steps", "1", "--no-cpu-baseline"], env=env,
                       capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")][0])
    rf = d["roofline"]
    assert rf["traffic_source"].startswith("measured in this run"), rf["traffic_source"]
    assert rf["traffic"] is not None and rf["traffic"] > 0, rf
    assert rf["frac_traffic"] is not None and 0 < rf["frac_traffic"] < 1.0
    regions = dict(rf["other_kernels"], **{rf["kernel"]: rf})
    for name, k in regions.items():
        if name.startswith(("tp_", "radial_mlp", "node_linear", "node_fused", "gate")):
            assert k["traffic"] is not None and k["traffic"] > 0, (name, k)
    for name, k in regions.items():
        if name.startswith("radial_mlp"):  # the MLP is priced on what it executes: 3 products per fp32 product
            products = 3.0  # (default modes: two-plane fp16 split in both directions)
            assert abs(k["frac_mfma"] - products * k["algorithmic_fp32_tflops"] / 2500.0) < 1e-9
            assert abs(k["frac_hbm"] - k["hbm_gbps"] / 8000.0) < 1e-9
            assert k["frac"] == max(k["frac_mfma"], k["frac_hbm"]) and k["bound"] in ("mfma", "hbm")
            assert k["peak"] == (2500.0 if k["bound"] == "mfma" else 8000.0)


def test_profile_summariser_classifies_every_generated_kernel():
    """scripts/summarize_profile.py maps GPU kernel names to bench.py's regions by *named template arguments*; every
    kernel instantiation the structure-specialised generator launches must land in a tensor-product region (CPU test:
    reads the generated sources)."""
    import glob
    import re

    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import summarize_profile as sp

    assert sp.region_of("void nqa::radial_mlp_fwd_split_bal_kernel<128, true>(float const*, float const*)") == ("radial_mlp_fwd", "main")
    assert sp.region_of("void nqa::radial_mlp_bwd_split_kernel<128, 0, false>(float const*)") == ("radial_mlp_bwd", "main")
    assert sp.region_of("nqa::radial_mlp_split_w1_fwd_f16_kernel(float const*)") == ("radial_mlp_fwd", "helper")
    spec_dir = os.path.join(ROOT, "nequip_amd", "csrc", "generated_spec")
    files = glob.glob(os.path.join(spec_dir, "*.hip"))
    if not files:  # (the generated sources are not tracked: make them, as build() would)
        sys.path.insert(0, os.path.join(ROOT, "nequip_amd", "csrc"))
        import gen_spec

        files = gen_spec.generate(spec_dir)
    seen = set()
    for f in files:
        for m in re.finditer(r"hipLaunchKernelGGL\(\((\w+<[^<>]*>)\)", open(f).read()):
            seen.add(m.group(1))
    assert len(seen) > 20
    want = {"fwd_kernel": {"tp_fwd"}, "bwd_x_kernel": {"tp_bwd_x"}, "bwd_edge_kernel": {"tp_bwd_edge", "tp_bwd_fused"},
            "bwd_pair_kernel": {"tp_bwd_edge", "tp_bwd_fused"}, "bwd_pair_split_kernel": {"tp_bwd_edge", "tp_bwd_fused"},
            "gx_rows_sum_kernel": {"tp_bwd_fused"}}
    for inst in sorted(seen):
        base = inst.split("<")[0]
        if base not in want:
            continue
        # the profiler prints `float` for T and literal values for the flags; FULL etc. are runtime-selected literals here
        name = "void nqa::(anonymous namespace)::" + inst + "(nqa::SpecArgs<float>)"
        region, _ = sp.region_of(name)
        assert region in want[base], (inst, region)
    # round 6: the LDS-ring pair kernel <WPN, GX, ATOM> (launched through a macro: not seen by the scan above) and its last step
    assert sp.region_of("void nqa::(anonymous namespace)::bwd_pair_ring_kernel<1, true, true>(nqa::SpecArgs<float>)") == ("tp_bwd_fused", "main")
    assert sp.region_of("void nqa::(anonymous namespace)::bwd_pair_ring_kernel<4, true, false>(nqa::SpecArgs<float>)") == ("tp_bwd_fused", "main")
    assert sp.region_of("void nqa::(anonymous namespace)::bwd_pair_ring_kernel<2, false, false>(nqa::SpecArgs<float>)") == ("tp_bwd_edge", "main")
    assert sp.region_of("nqa::(anonymous namespace)::gx_acc_finish_kernel(nqa::SpecArgs<float>)") == ("tp_bwd_fused", "helper")
    assert sp.region_of("void nqa::(anonymous namespace)::bwd_pair_split_kernel<float, true, true, true>(x)")[0] == "tp_bwd_fused"
    # the round-2 regression: a fifth template argument must not change the classification
    assert sp.region_of("void nqa::(anonymous namespace)::bwd_pair_kernel<float, 4, true, true, false>(x)")[0] == "tp_bwd_fused"
    assert sp.region_of("void nqa::(anonymous namespace)::bwd_pair_kernel<float, 4, true, false, true, 7>(x)")[0] == "tp_bwd_edge"


@pytest.mark.gpu
def test_bench_two_ranks_training_line_on_a_shared_device(device):
    """`bench.py --gpus 2 --workload train256` with both ranks on device 0 (gloo; a functional check of the N > 1 control
    flow, not a measurement): barriers, max over ranks, the flat gradient all-reduce inside the timed AND the
    kernel-instrumented steps (which every rank has to run: rank 0 alone used to wait for its peer forever), one JSON
    line from rank 0."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["NQA_BENCH_SHARE_DEVICE"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "train256",
                        "--steps", "2", "--warmup", "1", "--kernel-steps", "1", "--no-pmc", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=420)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "dp2" and "all-reduce" in d["config"]["collective"]
    # round 6: the gradient synchronisation timed on every rank (compute / collective split of a scaling curve)
    per_rank = d["config"]["collective_ms_per_rank"]
    assert isinstance(per_rank, list) and len(per_rank) == 2 and all(t > 0 for t in per_rank), per_rank
    assert d["value"] > 0 and "node_linear" in d["kernels_ms_per_step"]


def test_generated_kernel_sources_match_the_committed_manifest():
    """``csrc/generated_spec/*.hip`` (37 structure-specialised kernel files, ~100 k lines) are build products of
    ``gen_spec.py`` and are not tracked; ``csrc/generated_spec.manifest.json`` pins what the generator must emit (sha256 per
    file, default generator switches), so a change of the generator shows up as a one-line manifest diff in review, and
    whatever lies in ``generated_spec/`` (what the library was built from) must be exactly that."""
    import hashlib

    for k in list(os.environ):
        assert not k.startswith("NQA_GEN_"), f"{k} is set: the manifest pins the default generator switches"
    sys.path.insert(0, os.path.join(ROOT, "nequip_amd", "csrc"))
    import gen_spec

    want = json.load(open(gen_spec.MANIFEST))["files"]
    got = gen_spec.manifest()
    assert sorted(got) == sorted(want), (sorted(set(got) ^ set(want)))
    stale = [f for f in got if got[f]["sha256"] != want[f]["sha256"]]
    assert not stale, f"gen_spec.py output changed for {stale}: python nequip_amd/csrc/gen_spec.py --write-manifest"
    spec_dir = os.path.join(ROOT, "nequip_amd", "csrc", "generated_spec")
    if os.path.isdir(spec_dir):
        on_disk = {f for f in os.listdir(spec_dir) if f.endswith(".hip")}
        if on_disk:
            assert on_disk == set(want), sorted(on_disk ^ set(want))
            for f in sorted(on_disk):
                assert hashlib.sha256(open(os.path.join(spec_dir, f), "rb").read()).hexdigest() == want[f]["sha256"], f
