#!/bin/bash
# round 5, call 6: ablations of the issue-scheduled forward, LDS-epilogue form (never-taken branches; timing only)
OUT=gpurun_out/r5c6; mkdir -p $OUT
for d in 64 65 66 68 72 80 96 67 82 83 115 127; do
  echo "NQA_MLP_DBG=$d" | tee -a $OUT/mlp_ablate.log; NQA_MLP_PIPE_LDS=1 NQA_MLP_DBG=$d E=200279 FWD_ONLY=1 python scripts/bench_mlp.py 2>&1 | grep "H=" | tee -a $OUT/mlp_ablate.log
done
