#!/bin/bash
mkdir -p gpurun_out/r4ep
python -m pytest -x -q -m gpu tests/test_edge_pairs.py tests/test_topology_cache.py tests/test_ase_calculator.py tests/test_training_step.py > gpurun_out/r4ep/tests.log 2>&1
echo "rc=$?" >> gpurun_out/r4ep/tests.log
tail -3 gpurun_out/r4ep/tests.log | cut -c 1-600
for i in 1 2; do
python scripts/bench_md.py 2>&1 | tail -1
NQA_NO_EARLY_PAIRING=1 python scripts/bench_md.py 2>&1 | tail -1 | sed 's/^/[no early] /'
done
