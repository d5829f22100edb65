#!/bin/bash
# round 5, call 24: kernel stats + timeline of the cfg-4-shaped training step
cd /tmp && export TMPDIR=/tmp
D=$GRAFT_REPO_ROOT/gpurun_out/r5c24; rm -rf $D; mkdir -p $D
rocprofv3 --kernel-trace --stats --output-format csv -d $D/trace -o t -- python $GRAFT_REPO_ROOT/bench.py --workload train256 --steps 10 --warmup 3 --no-cpu-baseline --no-pmc > $D/bench.json 2> $D/bench.err
python - "$D" <<'PY'
import csv, glob, sys, collections
d = sys.argv[1]
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# steps are graph replays: split at the largest gaps is unreliable; take the last 30 % of the trace and aggregate per kernel name per step
n = len(rows)
tail = rows[int(n * 0.6):]
agg = collections.defaultdict(lambda: [0, 0.0])
for s, e, name in tail:
    short = name.replace("void ", "").replace("nqa::", "").replace("(anonymous namespace)::", "").split("(")[0][:90]
    agg[short][0] += 1; agg[short][1] += (e - s) / 1e3
span = (tail[-1][1] - tail[0][0]) / 1e3
tot = sum(v[1] for v in agg.values())
print("tail span us", round(span), "kernels", len(tail), "sum dur", round(tot))
with open(d + "/train_kernels.txt", "w") as fo:
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]:
        line = f"{v[1]/tot*100:5.1f}%  n={v[0]:5d}  avg {v[1]/v[0]:7.1f} us  {k}"
        print(line); fo.write(line + "\n")
PY
rm -rf $D/trace
