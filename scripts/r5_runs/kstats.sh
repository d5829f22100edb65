#!/bin/bash
# usage: kstats.sh <tag> [ENV=VAL ...] -- per-kernel average durations (rocprofv3 kernel trace) of scripts/bench_mlp.py
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
D=$GRAFT_REPO_ROOT/gpurun_out/kstats/$tag; rm -rf $D; mkdir -p $D
env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $D -o p -- python $GRAFT_REPO_ROOT/scripts/bench_mlp.py > $D/out.log 2>&1
python - "$D" "$tag" "$*" <<'PY'
import csv, glob, sys, collections
d, tag, envs = sys.argv[1], sys.argv[2], sys.argv[3]
rows = collections.defaultdict(list)
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "radial_mlp" not in n or "split_w1" in n: continue
        key = ("fwd" if "fwd" in n else "bwd") + ":" + n.split("<")[0].split("::")[-1] + "<" + n.split("<")[1].split(">")[0] + "> grid " + r.get("Grid_Size", r.get("Grid_Size_X", "?"))
        rows[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(rows.items()):
    v = sorted(v)
    # the benchmark alternates the two shapes; split by duration cluster: report quartiles
    print(f"[{tag}] {envs} | {k}: n={len(v)} min {v[0]:.1f} p25 {v[len(v)//4]:.1f} med {v[len(v)//2]:.1f} p75 {v[3*len(v)//4]:.1f} max {v[-1]:.1f} us")
PY
