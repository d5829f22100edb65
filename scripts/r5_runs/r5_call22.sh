#!/bin/bash
# round 5, call 22: does a HALF-chip persistent radial backward (1 workgroup per CU) disturb the main chain less?
OUT=gpurun_out/r5c22; mkdir -p $OUT
for rep in 1 2 3; do
for cfg in "NQA_X=0" "NQA_MLP_BWD_COAL=1" "NQA_MLP_BWD_COAL=1 NQA_MLP_BWD_WGS_PER_CU=1" "NQA_MLP_BWD_BALANCED=1 NQA_MLP_BWD_WGS_PER_CU=1"; do
  env $cfg python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-pmc > $OUT/b.json 2> $OUT/b.err
  python - "$cfg" <<'PY' | tee -a $OUT/ab.log
import json, sys
d = json.load(open("gpurun_out/r5c22/b.json"))
k = d["kernels_ms_per_step"]
print(sys.argv[1], "ms/step", round(d["ms_per_step"], 4), "mlp_bwd", round(k["radial_mlp_bwd"], 3), "node", round(k["node_linear"] + k["node_fused"], 3))
PY
done; done
