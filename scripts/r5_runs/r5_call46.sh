#!/bin/bash
mkdir -p gpurun_out/r5c46
timeout 900 python -m pytest tests/test_model_parity.py tests/test_edge_pairs.py -x -q -m gpu > gpurun_out/r5c46/tests.log 2>&1
tail -3 gpurun_out/r5c46/tests.log
for v in 1 0 1 0; do
  NQA_RADIAL_LAG=$v NQA_BENCH_NO_MD_STEP=1 NQA_BENCH_NO_EXACT_FP32=1 timeout 600 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-pmc --kernel-steps 0 2>>gpurun_out/r5c46/err.log | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('NQA_RADIAL_LAG=$v ms_per_step', round(r['ms_per_step'],4))" >> gpurun_out/r5c46/ab.log
done
cat gpurun_out/r5c46/ab.log
