#!/bin/bash
mkdir -p gpurun_out/r4own
python -m pytest -x -q -m gpu tests/test_edge_pairs.py tests/test_cpp_torch_ops.py tests/test_model_parity.py > gpurun_out/r4own/tests.log 2>&1
echo "rc=$?" >> gpurun_out/r4own/tests.log
tail -4 gpurun_out/r4own/tests.log | cut -c 1-800
python scripts/bench_topo.py 2>&1 | tail -7
python scripts/bench_md.py 2>&1 | tail -1
