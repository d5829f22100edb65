#!/bin/bash
mkdir -p gpurun_out/r5c33
timeout 900 python -m pytest tests/test_graphed_step.py -x -q -m gpu > gpurun_out/r5c33/tests.log 2>&1
tail -5 gpurun_out/r5c33/tests.log
bash scripts/r5_runs/timeline_md.sh > gpurun_out/r5c33/timeline.log 2>&1
cp gpurun_out/timeline_md/timeline.txt gpurun_out/r5c33/ 2>/dev/null
cp gpurun_out/timeline_md/bench.json gpurun_out/r5c33/md.log 2>/dev/null
rm -rf gpurun_out/timeline_md
grep -v "^ " gpurun_out/r5c33/timeline.log | tail -4
