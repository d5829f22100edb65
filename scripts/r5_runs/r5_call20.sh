#!/bin/bash
# round 5, call 20: embedding launches fused + backward seeded at the per-atom energies: parity, then whole-step A/B
OUT=gpurun_out/r5c20; mkdir -p $OUT
python -m pytest -x -q -m gpu tests/test_model_parity.py tests/test_converted_reference_model.py tests/test_edge_embed.py tests/test_golden_gpu.py tests/test_model_properties_gpu.py tests/test_traceable_model.py > $OUT/tests.log 2>&1; tail -3 $OUT/tests.log
for rep in 1 2 3; do
for cfg in "NQA_X=0" "NQA_NO_EMBED_FUSION=1 NQA_NO_ENERGY_SEED=1" ; do
  env $cfg python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-pmc > $OUT/b.json 2> $OUT/b.err
  python - "$cfg" <<'PY' | tee -a $OUT/ab.log
import json, sys
d = json.load(open("gpurun_out/r5c20/b.json"))
print(sys.argv[1], "ms/step", round(d["ms_per_step"], 4))
PY
done; done
