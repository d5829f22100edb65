#!/bin/bash
# round 5, call 16: backward with coalesced 256-byte row pieces + LDS transpose: parity, durations vs lane-=-row loads
OUT=gpurun_out/r5c16; mkdir -p $OUT
python -m pytest -x -q -m gpu tests/test_radial_mlp.py tests/test_edge_pairs.py tests/test_reference_golden.py > $OUT/tests.log 2>&1; tail -3 $OUT/tests.log
for rep in 1 2; do
for cfg in "NQA_MLP_PIPE=0" "NQA_MLP_BWD_COAL=0" "NQA_MLP_BWD_COAL=1"; do
for shape in 704 192; do
  bash scripts/r5_runs/kstats.sh b${shape}_$(echo $cfg | tr ' =' '__') $cfg SHAPES=$shape E=200279 2>&1 | grep "^\[" | grep bwd | tee -a $OUT/kstats.log
done; done; done
