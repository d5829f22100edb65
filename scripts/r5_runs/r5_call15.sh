#!/bin/bash
# round 5, call 15: full GPU suite + default bench at the commit that switches the radial MLP to radial_mlp_pipe.h
OUT=gpurun_out/r5c15; mkdir -p $OUT
python -m pytest tests -m gpu -q -x > $OUT/gpu_tests.log 2>&1; tail -4 $OUT/gpu_tests.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; python - <<'PY'
import json
d = json.load(open("gpurun_out/r5c15/bench.json"))
print("ms/step", round(d["ms_per_step"], 4), "value", round(d["value"]))
print({k: round(v, 3) for k, v in d["kernels_ms_per_step"].items()})
PY
NQA_MLP_PIPE=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_pipe0.json 2> $OUT/bench_pipe0.err; python - <<'PY'
import json
d = json.load(open("gpurun_out/r5c15/bench_pipe0.json"))
print("NQA_MLP_PIPE=0 ms/step", round(d["ms_per_step"], 4), "value", round(d["value"]))
print({k: round(v, 3) for k, v in d["kernels_ms_per_step"].items()})
PY
