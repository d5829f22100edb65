#!/bin/bash
# round 5, call 27: narrow-output resident backward ONLY for the launch that has the device to itself (first layer): A/B
OUT=gpurun_out/r5c27; mkdir -p $OUT
python -m pytest -x -q -m gpu tests/test_model_parity.py tests/test_baseline_size_parity.py -k "not cfg5_full and not cfg4" > $OUT/tests.log 2>&1; tail -2 $OUT/tests.log
for rep in 1 2 3 4; do
for cfg in "NQA_MLP_BWD_SMALL=0" "NQA_MLP_BWD_SMALL=1"; do
  env $cfg python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-pmc > $OUT/b.json 2> $OUT/b.err
  python - "$cfg" <<'PY' | tee -a $OUT/ab.log
import json, sys
d = json.load(open("gpurun_out/r5c27/b.json"))
k = d["kernels_ms_per_step"]
print(sys.argv[1], "ms/step", round(d["ms_per_step"], 4), "mlp_bwd", round(k["radial_mlp_bwd"], 3))
PY
done; done
