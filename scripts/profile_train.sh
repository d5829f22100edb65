#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel-trace stats of the cfg-4 training bench.
# usage: bash scripts/profile_train.sh <tag>
TAG=${1:-r1_train}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- \
  python $GRAFT_REPO_ROOT/bench.py --workload train256 --steps 10 --warmup 3 > $OUT/bench_trace.json 2> $OUT/bench_trace.err
rm -f $OUT/trace/bench_kernel_trace.csv
python $GRAFT_REPO_ROOT/scripts/summarize_profile.py $OUT $TAG > /dev/null
cat $OUT/bench_trace.json | cut -c1-300
