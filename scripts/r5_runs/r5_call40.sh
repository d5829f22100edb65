#!/bin/bash
# parameter-side stream in training: parity tests of the training path, then same-box A/B of the cfg-4-shaped step
mkdir -p gpurun_out/r5c40
timeout 1200 python -m pytest tests/test_training_step.py tests/test_mlp_training_fn.py tests/test_node_kernels.py tests/test_wgrad.py tests/test_baseline_size_parity.py -x -q -m gpu -k "not cfg5 and not cfg3" > gpurun_out/r5c40/tests.log 2>&1
tail -6 gpurun_out/r5c40/tests.log
for rep in 1 2; do
for v in 1 0; do
  NQA_PARAM_STREAM=$v timeout 600 python bench.py --workload train256 --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --kernel-steps 0 2>>gpurun_out/r5c40/err.log | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('NQA_PARAM_STREAM=$v ms_per_step', round(r['ms_per_step'],4), 'final_loss', r['config'].get('final_loss'))" >> gpurun_out/r5c40/ab.log
done; done
cat gpurun_out/r5c40/ab.log; tail -3 gpurun_out/r5c40/err.log
