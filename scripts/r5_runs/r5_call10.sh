#!/bin/bash
# round 5, call 10: balanced backward with deeper g_w prefetch: parity, kernel-trace durations old / new (bwd only shapes)
OUT=gpurun_out/r5c10; mkdir -p $OUT
python -m pytest -x -q -m gpu tests/test_radial_mlp.py tests/test_edge_pairs.py > $OUT/tests.log 2>&1; tail -2 $OUT/tests.log
for rep in 1 2; do
for shape in 704 192; do
for cfg in "NQA_MLP_PIPE=0" "NQA_MLP_PIPE=1"; do
  bash scripts/r5_runs/kstats.sh b${shape}_$(echo $cfg | tr ' =' '__') $cfg SHAPES=$shape E=200279 2>&1 | grep "^\[" | grep bwd | tee -a $OUT/kstats.log
done; done; done
