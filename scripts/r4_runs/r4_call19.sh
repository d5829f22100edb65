#!/bin/bash
mkdir -p gpurun_out/r4s
python -m pytest -x -q -m gpu tests/test_cpp_torch_ops.py > gpurun_out/r4s/tests_cpp.log 2>&1
echo "rc=$?" >> gpurun_out/r4s/tests_cpp.log
tail -30 gpurun_out/r4s/tests_cpp.log
python -m pytest -x -q -m gpu tests/test_traceable_model.py tests/test_aot_inductor.py tests/test_node_kernels.py tests/test_node_fused.py tests/test_topology_cache.py tests/test_reference_golden.py > gpurun_out/r4s/tests.log 2>&1
echo "rc=$?" >> gpurun_out/r4s/tests.log
tail -8 gpurun_out/r4s/tests.log
timeout 900 python scripts/bench_deployed.py > gpurun_out/r4s/deployed.log 2> gpurun_out/r4s/deployed.err
grep '^{' gpurun_out/r4s/deployed.log | cut -c 1-2500
