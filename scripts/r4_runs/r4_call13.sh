#!/bin/bash
mkdir -p gpurun_out/r4m
python -m pytest -x -q -m gpu tests/test_node_kernels.py tests/test_node_fused.py tests/test_training_step.py tests/test_model_parity.py > gpurun_out/r4m/tests.log 2>&1
echo "rc=$?" >> gpurun_out/r4m/tests.log
tail -8 gpurun_out/r4m/tests.log
for i in 1 2; do
  python bench.py --workload train256 --no-pmc --no-cpu-baseline > gpurun_out/r4m/train_order_$i.json 2> gpurun_out/r4m/train_order_$i.err
  NQA_NODE_TYPE_ORDER=0 python bench.py --workload train256 --no-pmc --no-cpu-baseline > gpurun_out/r4m/train_noorder_$i.json 2> gpurun_out/r4m/train_noorder_$i.err
done
python bench.py --no-pmc --no-cpu-baseline > gpurun_out/r4m/bench_order.json 2> gpurun_out/r4m/bench_order.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4m/*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][0])
    except Exception as e:
        print(f, 'ERR', e); continue
    print(f, round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['kernels_ms_per_step'].items() if k.startswith('node') or k.startswith('wgrad')})
PY
