#!/bin/bash
mkdir -p gpurun_out/r4w
python -m pytest -x -q -m gpu tests/test_traceable_model.py tests/test_training_step.py tests/test_node_kernels.py > gpurun_out/r4w/tests.log 2>&1
echo "rc=$?" >> gpurun_out/r4w/tests.log
tail -12 gpurun_out/r4w/tests.log | cut -c 1-1500
for i in 1 2; do
python bench.py --workload train256 --no-pmc --no-cpu-baseline > gpurun_out/r4w/train_$i.json 2> gpurun_out/r4w/train_$i.err
grep -o '"ms_per_step": [0-9.]*' gpurun_out/r4w/train_$i.json
done
