#!/bin/bash
# round 5, call 25: where does the W = 192 backward spend its 90 us?  (balanced kernel, ablation bits; 102 = no MFMA / staging / fragment reads)
OUT=gpurun_out/r5c25; mkdir -p $OUT
for cfg in "NQA_MLP_PIPE=0" "NQA_MLP_BWD_BALANCED=1" "NQA_MLP_BWD_BALANCED=1 NQA_MLP_DBG_BWD=64" "NQA_MLP_BWD_BALANCED=1 NQA_MLP_DBG_BWD=66" "NQA_MLP_BWD_BALANCED=1 NQA_MLP_DBG_BWD=102" "NQA_MLP_BWD_COAL=1"; do
  bash scripts/r5_runs/kstats.sh b192_$(echo $cfg | tr ' =' '__') $cfg SHAPES=192 E=200279 2>&1 | grep "^\[" | grep bwd | tee -a $OUT/kstats.log
done
