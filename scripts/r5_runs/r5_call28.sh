#!/bin/bash
# round 5, call 28: per-pair radial rows straight out of the embedding launch (no pair_gather / pair_expand): parity, A/B
OUT=gpurun_out/r5c28; mkdir -p $OUT
python -m pytest -x -q -m gpu tests/test_model_parity.py tests/test_converted_reference_model.py tests/test_edge_embed.py tests/test_golden_gpu.py tests/test_model_properties_gpu.py tests/test_full_size_properties.py tests/test_edge_pairs.py tests/test_topology_cache.py tests/test_ase_calculator.py > $OUT/tests.log 2>&1; tail -3 $OUT/tests.log
for rep in 1 2 3; do
for cfg in "NQA_X=0" "NQA_NO_PAIRED_EMBED=1" ; do
  env $cfg python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-pmc > $OUT/b.json 2> $OUT/b.err
  python - "$cfg" <<'PY' | tee -a $OUT/ab.log
import json, sys
d = json.load(open("gpurun_out/r5c28/b.json"))
print(sys.argv[1], "ms/step", round(d["ms_per_step"], 4))
PY
done; done
