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
    assert {"tp_fwd", "radial_mlp_fwd", "radial_mlp_bwd", "node_linear", "gate", "edge_embed_fwd", "edge_embed_bwd"} <= ran
    assert ran & {"tp_bwd_fused", "tp_bwd_edge"}
