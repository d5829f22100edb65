#!/bin/bash
# round 4, GPU call 5: fast activations in the fused node stage (parity), then the rocprofv3 passes of profile.sh
mkdir -p gpurun_out/r4e
python -m pytest -x -q -m gpu tests/test_node_fused.py tests/test_model_parity.py > gpurun_out/r4e/tests.log 2>&1
echo "rc=$?" >> gpurun_out/r4e/tests.log
tail -5 gpurun_out/r4e/tests.log
python bench.py --no-pmc --no-cpu-baseline > gpurun_out/r4e/bench_1.json 2> gpurun_out/r4e/bench_1.err
timeout 900 bash scripts/profile.sh r4 > gpurun_out/r4e/profile.log 2>&1
cp gpurun_out/prof_r4/r4_* gpurun_out/prof_r4/bench_trace.json gpurun_out/r4e/ 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4e/bench_*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][0])
        print(f, round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['kernels_ms_per_step'].items()})
    except Exception as e:
        print(f, 'ERR', e)
PY
