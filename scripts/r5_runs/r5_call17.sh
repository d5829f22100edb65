#!/bin/bash
# round 5, call 17: whole-step A/B of the radial-MLP kernel choices (same box, hipGraph replay, 3 repetitions each)
OUT=gpurun_out/r5c17; mkdir -p $OUT
python -m pytest -x -q -m gpu tests/test_radial_mlp.py tests/test_edge_pairs.py tests/test_model_parity.py > $OUT/tests.log 2>&1; tail -2 $OUT/tests.log
for rep in 1 2 3; do
for cfg in "NQA_MLP_PIPE=0" "NQA_MLP_PIPE=1" "NQA_MLP_BWD_COAL=0" ; do
  env $cfg python bench.py --steps 30 --warmup 5 --no-cpu-baseline > $OUT/b.json 2> $OUT/b.err
  python - "$cfg" <<'PY' | tee -a $OUT/ab.log
import json, sys
d = json.load(open("gpurun_out/r5c17/b.json"))
k = d["kernels_ms_per_step"]
print(sys.argv[1], "ms/step", round(d["ms_per_step"], 4), "mlp_fwd", round(k["radial_mlp_fwd"], 3), "mlp_bwd", round(k["radial_mlp_bwd"], 3), "tp_fwd", round(k["tp_fwd"], 3), "tp_bwd_fused", round(k["tp_bwd_fused"], 3))
PY
done; done
