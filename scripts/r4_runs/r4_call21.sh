#!/bin/bash
mkdir -p gpurun_out/r4u
python -m pytest -x -q -m gpu tests/test_edge_pairs.py tests/test_cpp_torch_ops.py tests/test_model_parity.py tests/test_baseline_size_parity.py tests/test_traceable_model.py > gpurun_out/r4u/tests.log 2>&1
echo "rc=$?" >> gpurun_out/r4u/tests.log
tail -12 gpurun_out/r4u/tests.log | cut -c 1-1500
python bench.py --no-pmc --no-cpu-baseline > gpurun_out/r4u/bench.json 2> gpurun_out/r4u/bench.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r4u/bench.json') if l.startswith('{')][0])
print('bench', round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['kernels_ms_per_step'].items() if k.startswith('tp')})
PY
python scripts/bench_md.py > gpurun_out/r4u/md.log 2>&1; tail -2 gpurun_out/r4u/md.log
NQA_NO_PAIRED=1 python scripts/bench_md.py > gpurun_out/r4u/md_nopair.log 2>&1; tail -1 gpurun_out/r4u/md_nopair.log
timeout 900 python scripts/bench_deployed.py > gpurun_out/r4u/deployed.log 2> gpurun_out/r4u/deployed.err
grep '^{' gpurun_out/r4u/deployed.log | cut -c 1-1200
