#!/bin/bash
# round 5, call 18: whole-step A/B: main chain captured on a high-priority stream; no overlap at all; default
OUT=gpurun_out/r5c18; mkdir -p $OUT
for rep in 1 2 3; do
for cfg in "NQA_X=0" "NQA_BENCH_MAIN_PRIO=1" "NQA_NO_OVERLAP=1" ; do
  env $cfg python bench.py --steps 30 --warmup 5 --no-cpu-baseline > $OUT/b.json 2> $OUT/b.err
  python - "$cfg" <<'PY' | tee -a $OUT/ab.log
import json, sys
d = json.load(open("gpurun_out/r5c18/b.json"))
k = d["kernels_ms_per_step"]
print(sys.argv[1], "ms/step", round(d["ms_per_step"], 4), "mlp_fwd", round(k["radial_mlp_fwd"], 3), "mlp_bwd", round(k["radial_mlp_bwd"], 3), "tp_fwd", round(k["tp_fwd"], 3), "tp_bwd_fused", round(k["tp_bwd_fused"], 3), "node", round(k["node_linear"] + k["node_fused"], 3))
PY
done; done
