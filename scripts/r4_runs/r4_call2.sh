#!/bin/bash
# round 4, GPU call 2: the fused node stage -- parity first, then same-box A/B of the cfg-3 step
mkdir -p gpurun_out/r4b
python -m pytest -x -q -m gpu -s tests/test_node_fused.py > gpurun_out/r4b/fused_tests.log 2>&1
echo "rc=$?" >> gpurun_out/r4b/fused_tests.log
tail -25 gpurun_out/r4b/fused_tests.log
python -m pytest -x -q -m gpu tests/test_node_kernels.py tests/test_model_parity.py tests/test_full_size_properties.py tests/test_ghost_exchange.py > gpurun_out/r4b/model_tests.log 2>&1
echo "rc=$?" >> gpurun_out/r4b/model_tests.log
tail -8 gpurun_out/r4b/model_tests.log
for i in 1 2; do
  python bench.py --no-pmc --no-cpu-baseline > gpurun_out/r4b/bench_fused_$i.json 2> gpurun_out/r4b/bench_fused_$i.err
  NQA_NO_NODE_FUSION=1 python bench.py --no-pmc --no-cpu-baseline > gpurun_out/r4b/bench_unfused_$i.json 2> gpurun_out/r4b/bench_unfused_$i.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4b/bench_*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][0])
        print(f, round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['kernels_ms_per_step'].items() if k.startswith(('node','gate'))})
    except Exception as e:
        print(f, 'ERR', e)
PY
