#!/bin/bash
# round 4, GPU call 8: the whole GPU suite + smoke, as the driver runs them
mkdir -p gpurun_out/r4h
python -m pytest tests/ -x -q -m gpu > gpurun_out/r4h/gpu_suite.log 2>&1
echo "rc=$?" >> gpurun_out/r4h/gpu_suite.log
tail -8 gpurun_out/r4h/gpu_suite.log
python __graft_entry__.py smoke > gpurun_out/r4h/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/r4h/smoke.log; tail -3 gpurun_out/r4h/smoke.log
