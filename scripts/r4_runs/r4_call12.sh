#!/bin/bash
mkdir -p gpurun_out/r4l
python -m pytest -x -q -m gpu tests/test_node_fused.py tests/test_model_parity.py tests/test_ghost_exchange.py tests/test_presets.py > gpurun_out/r4l/tests.log 2>&1
echo "rc=$?" >> gpurun_out/r4l/tests.log
tail -6 gpurun_out/r4l/tests.log
for i in 1 2; do
  python bench.py --no-pmc --no-cpu-baseline > gpurun_out/r4l/bench_order_$i.json 2> gpurun_out/r4l/bench_order_$i.err
  NQA_NODE_TYPE_ORDER=0 python bench.py --no-pmc --no-cpu-baseline > gpurun_out/r4l/bench_noorder_$i.json 2> gpurun_out/r4l/bench_noorder_$i.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4l/bench_*.json')):
    d=json.loads([l for l in open(f) if l.startswith('{')][0])
    print(f, round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['kernels_ms_per_step'].items() if k.startswith('node')})
PY
