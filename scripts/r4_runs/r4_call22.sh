#!/bin/bash
mkdir -p gpurun_out/r4v
python -m pytest -x -q -m gpu tests/test_edge_pairs.py tests/test_model_parity.py tests/test_presets.py tests/test_ase_calculator.py > gpurun_out/r4v/tests.log 2>&1
echo "rc=$?" >> gpurun_out/r4v/tests.log
tail -5 gpurun_out/r4v/tests.log | cut -c 1-1500
for i in 1 2; do
  python bench.py --no-pmc --no-cpu-baseline > gpurun_out/r4v/bench_prefetch_$i.json 2> gpurun_out/r4v/bench_prefetch_$i.err
  NQA_NO_RADIAL_PREFETCH=1 python bench.py --no-pmc --no-cpu-baseline > gpurun_out/r4v/bench_noprefetch_$i.json 2> gpurun_out/r4v/bench_noprefetch_$i.err
done
python bench.py --workload cu20k --no-pmc --no-cpu-baseline > gpurun_out/r4v/cu20k_prefetch.json 2> gpurun_out/r4v/cu20k_prefetch.err
NQA_NO_RADIAL_PREFETCH=1 python bench.py --workload cu20k --no-pmc --no-cpu-baseline > gpurun_out/r4v/cu20k_noprefetch.json 2> gpurun_out/r4v/cu20k_noprefetch.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4v/*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][0])
        print(f, round(d['ms_per_step'],4))
    except Exception as e:
        print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-600:])
PY
