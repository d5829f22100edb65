#!/bin/bash
# round 5, call 5: issue-scheduled forward with the direct (no LDS transpose) epilogue: parity, then A/B alone
OUT=gpurun_out/r5c5; mkdir -p $OUT
python -m pytest -x -q -m gpu tests/test_radial_mlp.py tests/test_reference_golden.py > $OUT/tests.log 2>&1; tail -3 $OUT/tests.log
for rep in 1 2; do
for cfg in "NQA_MLP_PIPE=0" "NQA_MLP_PIPE=1" "NQA_MLP_PIPE=1 NQA_MLP_PIPE_LDS=1"; do
  echo "$cfg" | tee -a $OUT/mlp_ab.log; env $cfg E=200279 FWD_ONLY=1 python scripts/bench_mlp.py 2>&1 | grep "H=" | tee -a $OUT/mlp_ab.log
done; done
