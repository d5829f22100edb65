#!/bin/bash
mkdir -p gpurun_out/r5c34
timeout 900 python -m pytest tests/test_graphed_step.py tests/test_neighbor_list.py tests/test_edge_pairs.py tests/test_topology_cache.py -x -q -m gpu > gpurun_out/r5c34/tests.log 2>&1
tail -5 gpurun_out/r5c34/tests.log
timeout 600 python scripts/bench_md.py > gpurun_out/r5c34/md.log 2>&1
tail -4 gpurun_out/r5c34/md.log
bash scripts/r5_runs/timeline_md.sh > gpurun_out/r5c34/timeline.log 2>&1
cp gpurun_out/timeline_md/timeline.txt gpurun_out/r5c34/ 2>/dev/null
rm -rf gpurun_out/timeline_md
grep -v "^ " gpurun_out/r5c34/timeline.log | tail -4
