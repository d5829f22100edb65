#!/bin/bash
mkdir -p gpurun_out/r4x
python -m pytest -q -m gpu tests > gpurun_out/r4x/tests_full.log 2>&1
echo "rc=$?" >> gpurun_out/r4x/tests_full.log
tail -6 gpurun_out/r4x/tests_full.log | cut -c 1-600
NQA_COMMIT=3d56e76 bash scripts/r4_collect.sh > gpurun_out/r4x/collect.log 2>&1
tail -6 gpurun_out/r4x/collect.log | cut -c 1-1500
