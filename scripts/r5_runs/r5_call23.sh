#!/bin/bash
# round 5, call 23: bench.py's N > 1 control flow on the one-GPU box (two ranks sharing device 0, gloo): functional check
OUT=gpurun_out/r5c23; mkdir -p $OUT
NQA_BENCH_SHARE_DEVICE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 10 --warmup 3 > $OUT/b2.json 2> $OUT/b2.err
tail -c 600 $OUT/b2.json; echo; tail -3 $OUT/b2.err | cut -c1-300
python bench.py --workload train256 --no-cpu-baseline --no-pmc 2>/dev/null | cut -c1-400
