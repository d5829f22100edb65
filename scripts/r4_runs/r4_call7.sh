#!/bin/bash
# round 4, GPU call 7: all-lanes (duplicate-lane) form of the backward TP kernels for every multiplicity -- parity, then A/B
mkdir -p gpurun_out/r4g
python -m pytest -x -q -m gpu tests/test_tp_spec_kernels.py tests/test_tp_scatter_kernel.py tests/test_presets.py tests/test_edge_pairs.py tests/test_training_step.py > gpurun_out/r4g/tests.log 2>&1
echo "rc=$?" >> gpurun_out/r4g/tests.log
tail -6 gpurun_out/r4g/tests.log
for w in water10k_S water10k_M water10k_L water10k_XL aspirin5; do
  timeout 400 python bench.py --workload $w --no-cpu-baseline --no-pmc 2> gpurun_out/r4g/$w.err >> gpurun_out/r4g/allanes.jsonl
  NQA_SPEC_MASKED=1 timeout 400 python bench.py --workload $w --no-cpu-baseline --no-pmc 2> gpurun_out/r4g/${w}_masked.err >> gpurun_out/r4g/masked.jsonl
done
python bench.py --no-pmc --no-cpu-baseline > gpurun_out/r4g/bench_1.json 2> gpurun_out/r4g/bench_1.err
python - <<'PY'
import json
for f in ('allanes','masked'):
    for l in open(f'gpurun_out/r4g/{f}.jsonl'):
        if not l.startswith('{'): continue
        d=json.loads(l)
        print(f, d['config']['workload'][:30], round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['kernels_ms_per_step'].items() if k.startswith('tp_')})
d=json.loads([l for l in open('gpurun_out/r4g/bench_1.json') if l.startswith('{')][0]); print('water10k', round(d['ms_per_step'],4))
PY
