#!/bin/bash
mkdir -p gpurun_out/r4o
python scripts/train_ops.py 3 > gpurun_out/r4o/train_ops.log 2>&1
python -m pytest -x -q -m gpu tests/test_node_kernels.py tests/test_training_step.py > gpurun_out/r4o/tests.log 2>&1
echo "rc=$?" >> gpurun_out/r4o/tests.log
tail -3 gpurun_out/r4o/tests.log
python bench.py --workload train256 --no-pmc --no-cpu-baseline > gpurun_out/r4o/train.json 2> gpurun_out/r4o/train.err
grep -o '"ms_per_step": [0-9.]*' gpurun_out/r4o/train.json
