set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/gputest.log
cat gpurun_out/gputest.log
timeout 600 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
tail -c 3000 gpurun_out/bench_default.json; tail -5 gpurun_out/bench_default.err
NQA_BENCH_SHARE_DEVICE=1 timeout 300 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline --no-pmc > gpurun_out/bench_n2_share.json 2> gpurun_out/bench_n2_share.err
tail -c 600 gpurun_out/bench_n2_share.json; tail -3 gpurun_out/bench_n2_share.err
