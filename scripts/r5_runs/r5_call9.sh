#!/bin/bash
# round 5, call 9: balanced f16x3 backward: parity tests, then kernel-trace durations old / new
OUT=gpurun_out/r5c9; mkdir -p $OUT
python -m pytest -x -q -m gpu tests/test_radial_mlp.py tests/test_reference_golden.py tests/test_edge_pairs.py > $OUT/tests.log 2>&1; tail -3 $OUT/tests.log
for shape in 704 192; do
for cfg in "NQA_MLP_PIPE=0" "NQA_MLP_PIPE=1"; do
  bash scripts/r5_runs/kstats.sh b${shape}_$(echo $cfg | tr ' =' '__') $cfg SHAPES=$shape E=200279 2>&1 | grep "^\[" | tee -a $OUT/kstats.log
done; done
