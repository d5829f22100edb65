#!/bin/bash
# round 5, call 3: issue-scheduled f16x3 radial-MLP forward (radial_mlp_pipe.h): parity tests, then A/B alone
OUT=gpurun_out/r5c3; mkdir -p $OUT
python -m pytest -x -q -m gpu tests/test_radial_mlp.py tests/test_reference_golden.py > $OUT/tests.log 2>&1; tail -5 $OUT/tests.log
for p in 0 1; do
  echo "NQA_MLP_PIPE=$p"; NQA_MLP_PIPE=$p E=200279 python scripts/bench_mlp.py 2>&1 | grep "H=" | tee -a $OUT/mlp_ab.log
done
