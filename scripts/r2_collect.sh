# End-of-round evidence: rocprofv3 kernel stats + PMC passes of the default bench (profiles/r2_*), the training bench
# (profiles/r2_train_*), the other BASELINE workloads (profiles/r2_other_workloads.jsonl) and the default bench line.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash scripts/profile.sh r2 > gpurun_out/profile_r2.log 2>&1
bash scripts/profile_train.sh r2_train > gpurun_out/profile_r2_train.log 2>&1
: > gpurun_out/r2_other_workloads.jsonl
for wl in aspirin5 si1k cu20k cu100k train256; do
  timeout 600 python bench.py --workload $wl --steps 10 --warmup 3 --no-pmc 2>/dev/null | tail -1 >> gpurun_out/r2_other_workloads.jsonl
done
timeout 600 python bench.py > gpurun_out/r2_bench_default.json 2> gpurun_out/r2_bench_default.err
tail -c 600 gpurun_out/r2_bench_default.json
