#!/bin/bash
mkdir -p gpurun_out/r5c39
bash scripts/r5_runs/timeline_train.sh > gpurun_out/r5c39/log.txt 2>&1
cp gpurun_out/timeline_train/tail.txt gpurun_out/r5c39/ 2>/dev/null
rm -rf gpurun_out/timeline_train
tail -3 gpurun_out/r5c39/log.txt
