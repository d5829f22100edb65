#!/bin/bash
mkdir -p gpurun_out/r4k
NQA_BENCH_SHARE_DEVICE=1 timeout 600 python bench.py --gpus 2 --steps 10 --warmup 3 --no-pmc > gpurun_out/r4k/bench_2ranks_shared.json 2> gpurun_out/r4k/bench_2ranks_shared.err
echo "rc=$?"; tail -c 600 gpurun_out/r4k/bench_2ranks_shared.json; tail -3 gpurun_out/r4k/bench_2ranks_shared.err
NQA_BENCH_SHARE_DEVICE=1 timeout 600 python bench.py --gpus 2 --workload train256 --steps 5 --warmup 2 --no-pmc > gpurun_out/r4k/train_2ranks_shared.json 2> gpurun_out/r4k/train_2ranks_shared.err
echo "rc=$?"; tail -c 400 gpurun_out/r4k/train_2ranks_shared.json; tail -3 gpurun_out/r4k/train_2ranks_shared.err
python -m pytest -q -m gpu tests/test_eval_mode_weights.py tests/test_model_parity.py 2>&1 | tail -3
