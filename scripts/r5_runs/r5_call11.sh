#!/bin/bash
# round 5, call 11: does the ragged last block cost the balanced backward 15 %?  (E = 200192 = 1564 full blocks vs 200279)
OUT=gpurun_out/r5c11; mkdir -p $OUT
for rep in 1 2; do
for E in 200192 200279 200320; do
  bash scripts/r5_runs/kstats.sh bE${E} NQA_MLP_PIPE=1 SHAPES=704 E=$E 2>&1 | grep "^\[" | grep bwd | tee -a $OUT/kstats.log
done; done
