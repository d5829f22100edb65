#!/bin/bash
mkdir -p gpurun_out/r4q
python -m pytest -x -q -m gpu tests/test_aot_inductor.py tests/test_ase_calculator.py tests/test_cpp_torch_ops.py > gpurun_out/r4q/tests.log 2>&1
echo "rc=$?" >> gpurun_out/r4q/tests.log
tail -15 gpurun_out/r4q/tests.log
python bench.py --no-graph --no-pmc --no-cpu-baseline > gpurun_out/r4q/bench_nograph.json 2> gpurun_out/r4q/bench_nograph.err
grep -o '"ms_per_step": [0-9.]*' gpurun_out/r4q/bench_nograph.json
timeout 900 python scripts/bench_deployed.py --profile > gpurun_out/r4q/deployed.log 2> gpurun_out/r4q/deployed.err
grep '^{' gpurun_out/r4q/deployed.log
grep -v Warning gpurun_out/r4q/deployed.err | head -80
