#!/usr/bin/env python3
"""Deployed forms of the cfg-3 evaluation against the eager model: time per energy + forces + virial
evaluation of (a) the eager model, (b) the make_fx graph run by Python, (c) the AOTInductor package through the Python
loader (ops registered from Python), all WITHOUT hipGraph replay (a deployed calculator launches its kernels every step),
plus the parity of each against (a) and the number of edge topologies built per evaluation.
Usage (GPU box): python scripts/bench_deployed.py [--workload water10k] [--steps 30] [--no-aoti]"""
import argparse
import json
import os
import sys
import tempfile
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from nequip_amd.data import AtomicDataDict  # noqa: E402
from nequip_amd.nn import _topology  # noqa: E402
from nequip_amd.utils.tracing import trace_model  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="water10k")
ap.add_argument("--steps", type=int, default=30)
ap.add_argument("--no-aoti", action="store_true")
ap.add_argument("--profile", action="store_true", help="cProfile of 10 eager evaluations (stderr)")
args = ap.parse_args()

dev = torch.device("cuda:0")
w = bench.WORKLOADS[args.workload]
data_cpu, names = bench.build_box(w, seed=0)
cfg = bench.model_cfg(w, data_cpu["edge_index"].shape[1] / data_cpu["pos"].shape[0])
model = bench.build_model(cfg, names, dev)
data = AtomicDataDict.to_device(data_cpu, dev)
FIELDS = ("pos", "edge_index", "atom_types", "cell", "edge_cell_shift")
inputs = {k: data[k] for k in FIELDS}
OUT = ("total_energy", "forces", "virial")

builds = [0]
_orig_init = _topology.EdgeTopology.__init__


def _counting_init(self, *a, **k):
    builds[0] += 1
    _orig_init(self, *a, **k)


_topology.EdgeTopology.__init__ = _counting_init


def timed(fn, label, ref=None):
    for _ in range(5):
        out = fn()
    torch.cuda.synchronize()
    builds[0] = 0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = fn()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / args.steps * 1e3
    rec = {"form": label, "ms_per_evaluation": round(ms, 4), "topologies_built_per_evaluation": builds[0] / args.steps}
    if ref is not None:
        for k, r, o in zip(OUT, ref, out):
            rec[f"max_abs_diff_{k}"] = float((r.double() - o.double()).abs().max())
    print(json.dumps(rec), flush=True)
    return out, ms


def pick(out):
    return [out[k].detach() for k in OUT]


ref, ms_eager = timed(lambda: pick(model(dict(inputs))), "eager model (no hipGraph)")
if args.profile:
    import cProfile
    import pstats

    pr = cProfile.Profile()
    pr.enable()
    for _ in range(10):
        model(dict(inputs))
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr, stream=sys.stderr).sort_stats("cumulative").print_stats(45)

t0 = time.time()
gm, params, buffers = trace_model(model, inputs, tracing_mode="symbolic")
ours = {}
for n in gm.graph.nodes:
    if n.op == "call_function":
        t = str(n.target)
        key = t.split(".")[1] if t.startswith("nequip_amd.") else "aten/other"
        ours[key] = ours.get(key, 0) + 1
print(json.dumps({"traced_in_s": round(time.time() - t0, 1), "graph_ops": ours}), flush=True)
_, ms_gm0 = timed(lambda: pick(gm(params, buffers, inputs)), "make_fx graph run by Python, weights as inputs", ref)
from nequip_amd.utils.tracing import fold_constants  # noqa: E402

print(json.dumps({"folded_buffers": fold_constants(gm, params, buffers)}), flush=True)
_, ms_gm = timed(lambda: pick(gm(params, buffers, inputs)), "make_fx graph run by Python, constants folded", ref)

if not args.no_aoti:
    from nequip_amd.utils.aot import aot_export_model, load_aotinductor_model

    path = os.path.join(tempfile.mkdtemp(), "cfg3.nequip.pt2")
    t0 = time.time()
    aot_export_model(model, data, path, input_fields=FIELDS, output_fields=OUT)
    print(json.dumps({"aoti_compiled_in_s": round(time.time() - t0, 1)}), flush=True)
    compiled, _ = load_aotinductor_model(path, device="cuda")
    _, ms_aoti = timed(lambda: pick(compiled(dict(inputs))), "AOTInductor package, Python-registered ops", ref)
    print(json.dumps({"aoti_over_eager": round(ms_aoti / ms_eager, 3), "graph_over_eager": round(ms_gm / ms_eager, 3)}),
          flush=True)
    # the package on the C++-registered ops alone (libnequip_amd_torch.so): a process that never imports nequip_amd
    import subprocess

    io = os.path.join(os.path.dirname(path), "io.pt")
    torch.save(([inputs[k].cpu() for k in FIELDS], [r.cpu() for r in ref]), io)
    child = r"""
import json, sys, time, torch
torch.ops.load_library(sys.argv[1])
assert "nequip_amd" not in sys.modules
compiled = torch._inductor.aoti_load_package(sys.argv[2])
inputs, ref = torch.load(sys.argv[3])
inputs = [t.cuda() for t in inputs]
steps = int(sys.argv[4])
for _ in range(5):
    out = compiled(inputs)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    out = compiled(inputs)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / steps * 1e3
rec = {"form": "AOTInductor package, C++-registered ops (topology cache mode " + sys.argv[5] + ")", "ms_per_evaluation": round(ms, 4)}
for k, r, o in zip(sys.argv[6:], ref, out):
    rec["max_abs_diff_" + k] = float((r.double() - o.cpu().double()).abs().max())
print(json.dumps(rec))
"""
    lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "nequip_amd", "csrc",
                       "libnequip_amd_torch.so")
    for mode in ("1", "2"):
        r = subprocess.run([sys.executable, "-c", child, lib, path, io, str(args.steps), mode] + list(OUT),
                           capture_output=True, text=True, timeout=900, cwd="/tmp",
                           env=dict(os.environ, PYTHONPATH="", NQA_TOPOLOGY_CACHE=mode))
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not line:
            print(json.dumps({"cpp_ops_run_failed": r.stderr[:1500] + " ... " + r.stderr[-600:]}), flush=True)
        else:
            rec = json.loads(line[-1])
            rec["over_eager"] = round(rec["ms_per_evaluation"] / ms_eager, 3)
            print(json.dumps(rec), flush=True)
else:
    print(json.dumps({"graph_over_eager": round(ms_gm / ms_eager, 3)}))
