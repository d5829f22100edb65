#!/bin/bash
mkdir -p gpurun_out/r5c41
timeout 900 python -m pytest tests/test_graphed_step.py tests/test_ase_calculator.py -x -q -m gpu > gpurun_out/r5c41/tests.log 2>&1
tail -4 gpurun_out/r5c41/tests.log
