#!/bin/bash
# usage (on the GPU box through gpurun): bash scripts/gpu_check.sh [bench args...]
mkdir -p gpurun_out
python -m pytest tests -m gpu -q 2>&1 | tail -n 120 > gpurun_out/test.log
grep -E "passed|failed|error" gpurun_out/test.log | tail -3
python bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" > gpurun_out/bench.json 2> gpurun_out/bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench.json"))
print("value", round(d["value"]), "ms/step", round(d["ms_per_step"], 3), d["config"]["launch"])
print("roofline", d["roofline"])
print("kernels ms/step", {k: round(v, 3) for k, v in d["kernels_ms_per_step"].items()})
print("kernels GB/s", {k: round(v) for k, v in d["kernels_gbps"].items()})
print("kernels TFLOP/s", {k: round(v, 1) for k, v in d.get("kernels_tflops", {}).items()})
PY
grep -v Warn gpurun_out/bench.err | grep -v amdgpu.ids | tail -5
