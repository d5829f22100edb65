#!/bin/bash
# the whole GPU suite at the final commit
mkdir -p gpurun_out/r5_final
timeout 1400 python -m pytest tests -q -m gpu > gpurun_out/r5_final/r5_gpu_tests_full.log 2>&1
tail -8 gpurun_out/r5_final/r5_gpu_tests_full.log
