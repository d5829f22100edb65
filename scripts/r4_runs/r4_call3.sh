#!/bin/bash
# round 4, GPU call 3: energy head + deep (depth >= 2) radial MLP on the HIP path
mkdir -p gpurun_out/r4c
python -m pytest -x -q -m gpu -s tests/test_node_fused.py tests/test_reference_golden.py "tests/test_baseline_size_parity.py::test_cfg1_tutorial_hyper_parameters_aspirin_batch" > gpurun_out/r4c/new_tests.log 2>&1
echo "rc=$?" >> gpurun_out/r4c/new_tests.log
tail -15 gpurun_out/r4c/new_tests.log
python -m pytest -x -q -m gpu tests/test_radial_mlp.py tests/test_model_parity.py tests/test_ghost_exchange.py tests/test_presets.py tests/test_model_properties_gpu.py > gpurun_out/r4c/model_tests.log 2>&1
echo "rc=$?" >> gpurun_out/r4c/model_tests.log
tail -6 gpurun_out/r4c/model_tests.log
python bench.py --no-pmc --no-cpu-baseline > gpurun_out/r4c/bench_1.json 2> gpurun_out/r4c/bench_1.err
NQA_NO_ENERGY_HEAD=1 python bench.py --no-pmc --no-cpu-baseline > gpurun_out/r4c/bench_nohead.json 2> gpurun_out/r4c/bench_nohead.err
python bench.py --no-pmc --no-cpu-baseline --workload aspirin5 > gpurun_out/r4c/bench_aspirin.json 2> gpurun_out/r4c/bench_aspirin.err
NQA_MLP_DEEP_ATEN=1 python bench.py --no-pmc --no-cpu-baseline --workload aspirin5 > gpurun_out/r4c/bench_aspirin_aten.json 2> gpurun_out/r4c/bench_aspirin_aten.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4c/bench_*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][0])
        print(f, round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['kernels_ms_per_step'].items()})
    except Exception as e:
        print(f, 'ERR', e)
PY
