#!/bin/bash
# kernel trace of the cfg-4-shaped training step (hipGraph replay of the optimizer step): the last ~3 steps as rows
cd /tmp && export TMPDIR=/tmp
D=$GRAFT_REPO_ROOT/gpurun_out/timeline_train; rm -rf $D; mkdir -p $D
env "$@" rocprofv3 --kernel-trace --output-format csv -d $D -o t -- python $GRAFT_REPO_ROOT/bench.py --workload train256 --steps 6 --warmup 2 --no-cpu-baseline --no-pmc --kernel-steps 0 > $D/bench.json 2> $D/bench.err
python - "$D" <<'PY'
import csv, glob, sys
d = sys.argv[1]
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
rows.sort()
rows = rows[-2600:]
t0 = rows[0][0]
with open(d + "/tail.txt", "w") as out:
    for s, e, n, q in rows:
        short = n.replace("void ", "").replace("nqa::", "").replace("(anonymous namespace)::", "").replace("at::native::", "")[:90]
        out.write(f"{(s - t0) / 1e3:10.1f} {(e - s) / 1e3:8.1f} q{q} {short}\n")
print(len(rows), "rows")
PY
tail -2 $D/bench.json | cut -c1-300
