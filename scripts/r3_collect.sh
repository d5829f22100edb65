#!/bin/bash
# Round-3 evidence in one gpurun call: the default bench line, the rocprofv3 passes of profile.sh / profile_train.sh and the
# bench lines of the other workloads, all stamped with NQA_COMMIT.  usage: NQA_COMMIT=<hash> bash scripts/r3_collect.sh
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3_final
mkdir -p $O
cd $R
echo "{\"commit\": \"${NQA_COMMIT:-n/a}\"}" > $O/commit.json
timeout 600 python bench.py > $O/r3_bench_default.json 2> $O/bench_default.err
timeout 600 bash scripts/profile.sh r3 > $O/profile.log 2>&1
timeout 300 bash scripts/profile_train.sh r3_train > $O/profile_train.log 2>&1
for w in si1k aspirin5 cu20k cu100k train256 water10k_S water10k_M water10k_L water10k_XL; do
  timeout 300 python bench.py --workload $w --no-cpu-baseline --no-pmc 2>/dev/null >> $O/r3_other_workloads.jsonl
done
cp $R/gpurun_out/prof_r3/r3_*  $R/gpurun_out/prof_r3/bench_trace.json $O/ 2>/dev/null
cp $R/gpurun_out/prof_r3/bench_trace_serial.json $O/bench_trace_serial.json 2>/dev/null
cp $R/gpurun_out/prof_r3_train/r3_train_kernel_stats_top40.csv $O/ 2>/dev/null
cp $R/gpurun_out/prof_r3_train/bench_trace.json $O/r3_train_bench_under_rocprof.json 2>/dev/null
ls $O
python - <<PY
import json
d = json.load(open("$O/r3_bench_default.json"))
print(d["ms_per_step"], d["value"], d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["traffic"], d["cpu_baseline"]["value"])
PY
