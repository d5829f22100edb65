#!/bin/bash
mkdir -p gpurun_out/r4nl
python -m pytest -x -q -m gpu tests/test_neighbor_list.py tests/test_ase_calculator.py tests/test_edge_pairs.py > gpurun_out/r4nl/tests.log 2>&1
echo "rc=$?" >> gpurun_out/r4nl/tests.log
tail -4 gpurun_out/r4nl/tests.log | cut -c 1-800
python scripts/bench_topo.py 2>&1 | tail -7
python scripts/bench_md.py 2>&1 | tail -1
python scripts/bench_nl.py 2>&1 | tail -6
