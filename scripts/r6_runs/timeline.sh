#!/bin/bash
# one hipGraph replay of the cfg-3 step as a timeline: start offset, duration, gap to the previous kernel's end, queue
cd /tmp && export TMPDIR=/tmp
D=$GRAFT_REPO_ROOT/gpurun_out/timeline; rm -rf $D; mkdir -p $D
env "$@" rocprofv3 --kernel-trace --output-format csv -d $D -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-pmc --kernel-steps 0 > $D/bench.json 2> $D/bench.err
python - "$D" <<'PY'
import csv, glob, sys
d = sys.argv[1]
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
rows.sort()
# the last complete step: find the last occurrence of the first kernel of a step (edge_vectors_fwd) and take from there
starts = [i for i, r in enumerate(rows) if "edge_vectors_fwd" in r[2]]
# the step with the shortest span (a hipGraph replay; eager steps carry host launch gaps)
best = None
for a, b in zip(starts, starts[1:]):
    span = max(r[1] for r in rows[a:b]) - rows[a][0]
    if best is None or span < best[0]:
        best = (span, a, b)
i0, i1 = best[1], best[2]
step = rows[i0:i1]
t0 = step[0][0]
out = open(d + "/timeline.txt", "w")
prev_end = t0
busy = 0
for s, e, n, q, st in step:
    short = n.replace("void ", "").replace("nqa::", "").replace("(anonymous namespace)::", "").split("(")[0][:70]
    line = f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  gap {(s - prev_end) / 1e3:6.1f}  q{q}/s{st}  {short}"
    out.write(line + "\n")
    prev_end = max(prev_end, e)
print("step span", (max(r[1] for r in step) - t0) / 1e3, "us;", len(step), "kernels; sum of durations", sum(e - s for s, e, *_ in step) / 1e3)
gaps = 0; pe = t0
for s, e, *_ in step:
    if s > pe: gaps += s - pe
    pe = max(pe, e)
print("idle (no kernel running) inside the step:", gaps / 1e3, "us")
PY
cat $D/timeline.txt | head -70
