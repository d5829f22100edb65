#!/bin/bash
# round 4, GPU call 1: the new parity tests + the tests whose checks gained an oracle leg
mkdir -p gpurun_out/r4a
free -g > gpurun_out/r4a/mem.txt; nproc >> gpurun_out/r4a/mem.txt
python -m pytest -x -q -m gpu -s \
  tests/test_baseline_size_parity.py tests/test_aot_inductor.py tests/test_ase_calculator.py tests/test_ghost_exchange.py \
  tests/test_cpp_torch_ops.py tests/test_presets.py > gpurun_out/r4a/tests.log 2>&1
echo "rc=$?" >> gpurun_out/r4a/tests.log
tail -5 gpurun_out/r4a/tests.log
