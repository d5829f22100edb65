#!/bin/bash
# round 5, call 4: ablations of the issue-scheduled forward (wrong results by construction; timing only)
OUT=gpurun_out/r5c4; mkdir -p $OUT
for d in 0 1 2 3 4 8 12 5 7 15; do
  echo "NQA_MLP_DBG=$d"; NQA_MLP_DBG=$d E=200279 FWD_ONLY=1 python scripts/bench_mlp.py 2>&1 | grep "H=" | tee -a $OUT/mlp_ablate.log
done
