#!/bin/bash
mkdir -p gpurun_out/r5c37
timeout 900 python -m pytest tests/test_ase_calculator.py tests/test_graphed_step.py -x -q -m gpu > gpurun_out/r5c37/tests.log 2>&1
tail -15 gpurun_out/r5c37/tests.log
