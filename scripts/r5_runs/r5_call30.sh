#!/bin/bash
# padded neighbour list / graphed MD step: new tests + the suites they touch, then the MD-like timing
mkdir -p gpurun_out/r5c30
timeout 900 python -m pytest tests/test_graphed_step.py tests/test_neighbor_list.py tests/test_edge_pairs.py tests/test_topology_cache.py -x -q -m gpu > gpurun_out/r5c30/tests.log 2>&1
tail -15 gpurun_out/r5c30/tests.log
timeout 600 python scripts/bench_md.py > gpurun_out/r5c30/md.log 2>&1
tail -12 gpurun_out/r5c30/md.log
