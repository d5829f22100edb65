#!/bin/bash
mkdir -p gpurun_out/r4j
python -m pytest -q -m gpu tests/test_eval_mode_weights.py tests/test_node_fused.py > gpurun_out/r4j/tests.log 2>&1
echo "rc=$?" >> gpurun_out/r4j/tests.log
tail -4 gpurun_out/r4j/tests.log
NQA_COMMIT=2219f10 bash scripts/r4_collect.sh > gpurun_out/r4j/collect.log 2>&1
tail -5 gpurun_out/r4j/collect.log
