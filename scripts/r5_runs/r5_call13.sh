#!/bin/bash
# round 5, call 13: ablations of the balanced backward (2 = no MFMA, 4 = no weight staging, 32 = no LDS fragment reads)
OUT=gpurun_out/r5c13; mkdir -p $OUT
for d in 64 66 68 96 70 102 38; do
  bash scripts/r5_runs/kstats.sh bd$d NQA_MLP_DBG_BWD=$d SHAPES=704 E=200279 2>&1 | grep "^\[" | grep bwd | tee -a $OUT/kstats.log
done
