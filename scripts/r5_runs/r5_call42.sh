#!/bin/bash
# deferred parameter gradients: parity, then same-box A/B of the cfg-4-shaped step
mkdir -p gpurun_out/r5c42
timeout 900 python -m pytest tests/test_training_step.py -x -q -m gpu -k "deferred or float32" > gpurun_out/r5c42/tests.log 2>&1
tail -12 gpurun_out/r5c42/tests.log
for v in 1 0 1 0; do
  NQA_DEFER_PARAM_GRADS=$v timeout 600 python bench.py --workload train256 --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --kernel-steps 0 2>>gpurun_out/r5c42/err.log | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('NQA_DEFER_PARAM_GRADS=$v ms_per_step', round(r['ms_per_step'],4), 'final_loss', r['config'].get('final_loss'))" >> gpurun_out/r5c42/ab.log
done
cat gpurun_out/r5c42/ab.log; grep -v "amdgpu.ids" gpurun_out/r5c42/err.log | tail -5
