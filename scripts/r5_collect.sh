#!/bin/bash
# Round-5 evidence in one gpurun call: the default bench line (live PMC traffic, CPU baseline), the rocprofv3 passes of
# profile.sh, a kernel timeline of one replayed step, the bench lines of the other workloads, the MD-like step.
# usage: NQA_COMMIT=<hash> bash scripts/r5_collect.sh
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5_final
mkdir -p $O
cd $R
echo "{\"commit\": \"${NQA_COMMIT:-n/a}\"}" > $O/r5_commit.json
timeout 1200 python bench.py > $O/r5_bench_default.json 2> $O/bench_default.err
timeout 900 bash scripts/profile.sh r5 > $O/profile.log 2>&1
timeout 300 bash scripts/r5_runs/timeline.sh > $O/timeline.log 2>&1
cp $R/gpurun_out/timeline/timeline.txt $O/r5_timeline_step.txt 2>/dev/null
tail -3 $O/timeline.log | head -2 >> $O/r5_timeline_step.txt
rm -f $O/r5_other_workloads.jsonl
for w in si1k aspirin5 cu20k cu100k train256; do
  timeout 400 python bench.py --workload $w --no-cpu-baseline --no-pmc 2>/dev/null >> $O/r5_other_workloads.jsonl
done
(echo "# scripts/bench_md.py, cfg-3 box, new neighbour list every step (eager launches)"; python scripts/bench_md.py 2>/dev/null | tail -1) > $O/r5_md_like_step.log
cp $R/gpurun_out/prof_r5/r5_* $O/ 2>/dev/null
cp $R/gpurun_out/prof_r5/bench_trace.json $O/r5_bench_under_rocprof.json 2>/dev/null
cp $R/gpurun_out/prof_r5/bench_trace_serial.json $O/r5_bench_under_rocprof_serial.json 2>/dev/null
ls $O
python - <<PY
import json
d = json.load(open("$O/r5_bench_default.json"))
print(d["ms_per_step"], d["value"], d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["traffic"], d["cpu_baseline"])
PY
