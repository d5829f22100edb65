#!/bin/bash
mkdir -p gpurun_out/r4x3
python -m pytest -q -m gpu tests > gpurun_out/r4x3/tests_full.log 2>&1
echo "rc=$?" >> gpurun_out/r4x3/tests_full.log
tail -4 gpurun_out/r4x3/tests_full.log | cut -c 1-600
NQA_COMMIT=94d0a56 bash scripts/r4_collect.sh > gpurun_out/r4x3/collect.log 2>&1
tail -3 gpurun_out/r4x3/collect.log | cut -c 1-900
python scripts/bench_topo.py 2>&1 | tail -7 > gpurun_out/r4_final/r4_topology_setup.log; cat gpurun_out/r4_final/r4_topology_setup.log
