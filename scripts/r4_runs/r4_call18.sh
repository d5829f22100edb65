#!/bin/bash
mkdir -p gpurun_out/r4r
python -m pytest -x -q -m gpu tests/test_cpp_torch_ops.py tests/test_traceable_model.py tests/test_aot_inductor.py tests/test_node_kernels.py tests/test_node_fused.py > gpurun_out/r4r/tests.log 2>&1
echo "rc=$?" >> gpurun_out/r4r/tests.log
tail -25 gpurun_out/r4r/tests.log
timeout 900 python scripts/bench_deployed.py > gpurun_out/r4r/deployed.log 2> gpurun_out/r4r/deployed.err
grep '^{' gpurun_out/r4r/deployed.log
grep -v Warning gpurun_out/r4r/deployed.err | tail -15
