#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel-trace stats of the default bench command, plus separate
# PMC passes (FETCH_SIZE / WRITE_SIZE each in their own run, as the MI355X guide prescribes).
# usage: bash scripts/profile.sh <tag>
TAG=${1:-r1}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- \
  python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-pmc > $OUT/bench_trace.json 2> $OUT/bench_trace.err
# the same command with the side-stream branches serialised: per-kernel durations that are comparable with bench.py's
# HIP-event timings (concurrent kernels stretch each other in the default trace)
NQA_NO_OVERLAP=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_serial -o bench -- \
  python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-pmc > $OUT/bench_trace_serial.json 2> $OUT/bench_trace_serial.err
rm -f $OUT/trace_serial/bench_kernel_trace.csv
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o bench -- \
  python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --kernel-steps 0 > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o bench -- \
  python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --kernel-steps 0 > /dev/null 2> $OUT/pmc_write.err
ls -R $OUT | head -30
# keep the merged output small: drop the raw per-dispatch trace, keep stats + counters
rm -f $OUT/trace/bench_kernel_trace.csv
python $GRAFT_REPO_ROOT/scripts/summarize_profile.py $OUT $TAG
