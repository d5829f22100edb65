#!/bin/bash
# round 5, call 14: balanced backward, g_w in two-chunk bursts (PF=4) vs one chunk per body (PF=2): parity + durations
OUT=gpurun_out/r5c14; mkdir -p $OUT
NQA_MLP_BWD_PF=4 python -m pytest -x -q -m gpu tests/test_radial_mlp.py tests/test_edge_pairs.py > $OUT/tests.log 2>&1; tail -2 $OUT/tests.log
for rep in 1 2; do
for cfg in "NQA_MLP_PIPE=0" "NQA_MLP_BWD_PF=2" "NQA_MLP_BWD_PF=4"; do
for shape in 704 192; do
  bash scripts/r5_runs/kstats.sh b${shape}_$(echo $cfg | tr ' =' '__') $cfg SHAPES=$shape E=200279 2>&1 | grep "^\[" | grep bwd | tee -a $OUT/kstats.log
done; done; done
