#!/bin/bash
# Round-6 evidence in one gpurun call: the default bench line (live PMC traffic, CPU baseline), the rocprofv3 passes of
# profile.sh, a kernel timeline of one replayed step, the bench lines of the other workloads, the MD-like step.
# usage: NQA_COMMIT=<hash> bash scripts/r6_collect.sh
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6_final
mkdir -p $O
cd $R
echo "{\"commit\": \"${NQA_COMMIT:-n/a}\"}" > $O/r6_commit.json
timeout 1200 python bench.py > $O/r6_bench_default.json 2> $O/bench_default.err
timeout 900 bash scripts/profile.sh r6 > $O/profile.log 2>&1
timeout 300 bash scripts/r6_runs/timeline.sh > $O/timeline.log 2>&1
cp $R/gpurun_out/timeline/timeline.txt $O/r6_timeline_step.txt 2>/dev/null
tail -3 $O/timeline.log | head -2 >> $O/r6_timeline_step.txt
rm -f $O/r6_other_workloads.jsonl
for w in si1k aspirin5 cu20k cu100k train256; do
  timeout 400 python bench.py --workload $w --no-cpu-baseline --no-pmc 2>/dev/null >> $O/r6_other_workloads.jsonl
done
(echo "# scripts/bench_md.py, cfg-3 box, new neighbour list every step (eager launches)"; python scripts/bench_md.py 2>/dev/null | tail -1) > $O/r6_md_like_step.log
cp $R/gpurun_out/prof_r6/r6_* $O/ 2>/dev/null
cp $R/gpurun_out/prof_r6/bench_trace.json $O/r6_bench_under_rocprof.json 2>/dev/null
cp $R/gpurun_out/prof_r6/bench_trace_serial.json $O/r6_bench_under_rocprof_serial.json 2>/dev/null
ls $O
python - <<PY
import json
d = json.load(open("$O/r6_bench_default.json"))
print(d["ms_per_step"], d["value"], d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["traffic"], d["cpu_baseline"])
PY
# cu100k timeline (kernel list of one step)
cd /tmp && export TMPDIR=/tmp
D=$R/gpurun_out/timeline_cu100k; rm -rf $D; mkdir -p $D
rocprofv3 --kernel-trace --stats --output-format csv -d $D -o t -- python $R/bench.py --workload cu100k --steps 3 --warmup 1 --no-cpu-baseline --no-pmc --kernel-steps 0 > $D/bench.json 2> $D/bench.err
python - "$D" > $O/r6_cu100k_kernel_stats.txt <<'PY'
import csv, glob, sys
d = sys.argv[1]
for f in glob.glob(d + "/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    for r in rows[:30]:
        print(f'{float(r["TotalDurationNs"])/1e6:9.2f} ms {100*float(r["TotalDurationNs"])/tot:5.1f}%  n={r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:9.1f} us  {r["Name"][:110]}')
PY
rm -rf $D
cat $O/r6_cu100k_kernel_stats.txt | head -30
