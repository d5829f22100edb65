#!/bin/bash
mkdir -p gpurun_out/r5c43
bash scripts/r5_runs/timeline_train.sh NQA_DEFER_PARAM_GRADS=1 > gpurun_out/r5c43/log.txt 2>&1
cp gpurun_out/timeline_train/tail.txt gpurun_out/r5c43/ 2>/dev/null
rm -rf gpurun_out/timeline_train
tail -3 gpurun_out/r5c43/log.txt | cut -c1-200
