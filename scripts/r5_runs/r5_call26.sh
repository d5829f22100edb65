#!/bin/bash
# round 5, call 26: LDS-resident backward for narrow outputs: parity, durations alone, whole-step A/B
OUT=gpurun_out/r5c26; mkdir -p $OUT
python -m pytest -x -q -m gpu tests/test_radial_mlp.py tests/test_edge_pairs.py tests/test_reference_golden.py tests/test_model_parity.py > $OUT/tests.log 2>&1; tail -3 $OUT/tests.log
for cfg in "NQA_MLP_BWD_SMALL=0" "NQA_MLP_BWD_SMALL=1"; do
  bash scripts/r5_runs/kstats.sh b192_$(echo $cfg | tr ' =' '__') $cfg SHAPES=192 E=200279 2>&1 | grep "^\[" | grep bwd | tee -a $OUT/kstats.log
done
for rep in 1 2 3; do
for cfg in "NQA_MLP_BWD_SMALL=0" "NQA_MLP_BWD_SMALL=1"; do
  env $cfg python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-pmc > $OUT/b.json 2> $OUT/b.err
  python - "$cfg" <<'PY' | tee -a $OUT/ab.log
import json, sys
d = json.load(open("gpurun_out/r5c26/b.json"))
k = d["kernels_ms_per_step"]
print(sys.argv[1], "ms/step", round(d["ms_per_step"], 4), "mlp_bwd", round(k["radial_mlp_bwd"], 3))
PY
done; done
