#!/bin/bash
# round 5, call 7: single-component ablations (everything off but one stream)
OUT=gpurun_out/r5c7; mkdir -p $OUT
for d in 127 126 125 123 119 111 95 124 122 64; do
  echo "NQA_MLP_DBG=$d" | tee -a $OUT/mlp_ablate.log; NQA_MLP_PIPE_LDS=1 NQA_MLP_DBG=$d E=200192 FWD_ONLY=1 python scripts/bench_mlp.py 2>&1 | grep "H=" | tee -a $OUT/mlp_ablate.log
done
