#!/bin/bash
mkdir -p gpurun_out/r4p
python -m pytest -x -q -m gpu tests/test_traceable_model.py tests/test_topology_cache.py tests/test_aot_inductor.py tests/test_ase_calculator.py > gpurun_out/r4p/tests.log 2>&1
echo "rc=$?" >> gpurun_out/r4p/tests.log
tail -15 gpurun_out/r4p/tests.log
timeout 900 python scripts/bench_deployed.py > gpurun_out/r4p/deployed.log 2> gpurun_out/r4p/deployed.err
grep '^{' gpurun_out/r4p/deployed.log
tail -5 gpurun_out/r4p/deployed.err
