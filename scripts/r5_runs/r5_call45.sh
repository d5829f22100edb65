#!/bin/bash
mkdir -p gpurun_out/r5c45
for v in "1 1" "0 0" "1 1" "0 0" "1 1" "0 0"; do
  set -- $v
  NQA_DEFER_PARAM_GRADS=$1 NQA_DEFER_LAG=$2 timeout 600 python bench.py --workload train256 --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --kernel-steps 0 2>>gpurun_out/r5c45/err.log | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('DEFER=$1 LAG=$2 ms_per_step', round(r['ms_per_step'],4), 'final_loss', r['config'].get('final_loss'))" >> gpurun_out/r5c45/ab.log
done
cat gpurun_out/r5c45/ab.log
