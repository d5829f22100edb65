#!/bin/bash
mkdir -p gpurun_out/r4t
python -m pytest -x -q -m gpu tests/test_cpp_torch_ops.py > gpurun_out/r4t/tests_cpp.log 2>&1
echo "rc=$?" >> gpurun_out/r4t/tests_cpp.log
tail -30 gpurun_out/r4t/tests_cpp.log | cut -c 1-1500
timeout 900 python scripts/bench_deployed.py > gpurun_out/r4t/deployed.log 2> gpurun_out/r4t/deployed.err
grep '^{' gpurun_out/r4t/deployed.log | cut -c 1-2500
