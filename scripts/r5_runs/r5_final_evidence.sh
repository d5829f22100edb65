#!/bin/bash
# evidence at the final commit, in the GPU time that is left: (1) the default bench line (live PMC traffic, CPU baseline on the
# full box, graphed MD step), (2) rocprofv3 --kernel-trace --stats of the same command, (3) a kernel timeline of one replayed step
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5_final
mkdir -p $O
cd $R
echo "{\"commit\": \"${NQA_COMMIT:-n/a}\"}" > $O/r5_commit.json
timeout 250 python bench.py > $O/r5_bench_default.json 2> $O/bench_default.err
python - <<PY
import json
try:
    d = json.loads(open("$O/r5_bench_default.json").read().strip().splitlines()[-1])
    print(d["ms_per_step"], d["value"], d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["traffic"], d.get("cpu_baseline", {}).get("full_box_value"), d["config"]["md_step"])
except Exception as e:
    print("bench line unreadable:", e)
PY
cd /tmp && export TMPDIR=/tmp
timeout 70 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- \
  env NQA_BENCH_NO_MD_STEP=1 NQA_BENCH_NO_EXACT_FP32=1 python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-pmc > $O/r5_bench_under_rocprof.json 2> $O/bench_trace.err
python - <<PY
import csv, glob
for f in glob.glob("$O/trace/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.reader(open(f)))
    with open("$O/r5_kernel_stats_top40.csv", "w", newline="") as out:
        csv.writer(out).writerows(rows[:41])
    print("kernel stats rows", len(rows))
PY
rm -rf $O/trace
cd $R
timeout 60 bash scripts/r5_runs/timeline.sh NQA_BENCH_NO_MD_STEP=1 NQA_BENCH_NO_EXACT_FP32=1 > $O/timeline.log 2>&1
cp $R/gpurun_out/timeline/timeline.txt $O/r5_timeline_step.txt 2>/dev/null
grep -v "^ " $O/timeline.log | tail -3 | head -2 >> $O/r5_timeline_step.txt
rm -rf $R/gpurun_out/timeline
ls $O
