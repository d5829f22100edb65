#!/bin/bash
# round 5, call 8: kernel-trace durations (no host time) of the radial-MLP kernels, one shape per run
OUT=gpurun_out/r5c8; mkdir -p $OUT
for shape in 704 192; do
for cfg in "NQA_MLP_PIPE=0" "NQA_MLP_PIPE=1" "NQA_MLP_PIPE=1 NQA_MLP_PIPE_LDS=1" "NQA_MLP_PIPE_LDS=1 NQA_MLP_DBG=127" "NQA_MLP_PIPE_LDS=1 NQA_MLP_DBG=126" "NQA_MLP_PIPE_LDS=1 NQA_MLP_DBG=125" "NQA_MLP_PIPE_LDS=1 NQA_MLP_DBG=124" "NQA_MLP_PIPE_LDS=1 NQA_MLP_DBG=111"; do
  bash scripts/r5_runs/kstats.sh s${shape}_$(echo $cfg | tr ' =' '__') $cfg SHAPES=$shape E=200192 FWD_ONLY=1 2>&1 | grep "^\[" | tee -a $OUT/kstats.log
done; done
