#!/bin/bash
# Round-4 evidence in one gpurun call: the default bench line (with live PMC traffic and the CPU baseline), the rocprofv3
# passes of profile.sh / profile_train.sh and the bench lines of the other workloads, all stamped with NQA_COMMIT.
# usage: NQA_COMMIT=<hash> bash scripts/r4_collect.sh
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4_final
mkdir -p $O
cd $R
echo "{\"commit\": \"${NQA_COMMIT:-n/a}\"}" > $O/r4_commit.json
timeout 900 python bench.py > $O/r4_bench_default.json 2> $O/bench_default.err
timeout 900 bash scripts/profile.sh r4 > $O/profile.log 2>&1
timeout 400 bash scripts/profile_train.sh r4_train > $O/profile_train.log 2>&1
rm -f $O/r4_other_workloads.jsonl
for w in si1k aspirin5 cu20k cu100k train256 water10k_S water10k_M water10k_L water10k_XL; do
  timeout 400 python bench.py --workload $w --no-cpu-baseline --no-pmc 2>/dev/null >> $O/r4_other_workloads.jsonl
done
timeout 600 python scripts/bench_deployed.py 2> $O/deployed.err | grep '^{' > $O/r4_deployed_forms.jsonl
(echo "# scripts/bench_md.py, cfg-3 box, new neighbour list every step (eager launches)"; python scripts/bench_md.py 2>/dev/null | tail -1;
 echo "# NQA_NO_PAIRED=1"; NQA_NO_PAIRED=1 python scripts/bench_md.py 2>/dev/null | tail -1) > $O/r4_md_like_step.log
cp $R/gpurun_out/prof_r4/r4_* $O/ 2>/dev/null
cp $R/gpurun_out/prof_r4/bench_trace.json $O/r4_bench_under_rocprof.json 2>/dev/null
cp $R/gpurun_out/prof_r4/bench_trace_serial.json $O/r4_bench_under_rocprof_serial.json 2>/dev/null
cp $R/gpurun_out/prof_r4_train/r4_train_kernel_stats_top40.csv $O/ 2>/dev/null
ls $O
python - <<PY
import json
d = json.load(open("$O/r4_bench_default.json"))
print(d["ms_per_step"], d["value"], d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["traffic"], d["cpu_baseline"])
PY
