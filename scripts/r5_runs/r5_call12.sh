#!/bin/bash
# round 5, call 12: balanced backward variants (prefetch depth 2 / 4, ragged-block code in or out), same box
OUT=gpurun_out/r5c12; mkdir -p $OUT
for rep in 1 2; do
for cfg in "NQA_MLP_PIPE=0 E=200192" "NQA_MLP_BWD_PF=2 E=200192" "NQA_MLP_BWD_PF=2 E=200279" "NQA_MLP_BWD_PF=4 E=200192" "NQA_MLP_BWD_PF=4 E=200279"; do
  bash scripts/r5_runs/kstats.sh b_$(echo $cfg | tr ' =' '__') $cfg SHAPES=704 2>&1 | grep "^\[" | grep bwd | tee -a $OUT/kstats.log
done; done
