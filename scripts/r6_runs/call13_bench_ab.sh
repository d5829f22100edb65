#!/bin/bash
# round 6, call 13: cfg-3 step with the three forms of the pair-centric backward (same box)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6_call13; rm -rf $OUT; mkdir -p $OUT
B="python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-pmc --no-other-workloads"
for rep in 1 2; do
NQA_PAIR_RING=0 $B > $OUT/registers_$rep.json 2> $OUT/registers_$rep.err
NQA_PAIR_GX_ATOMIC=0 $B > $OUT/ring_rows_$rep.json 2> $OUT/ring_rows_$rep.err
$B > $OUT/ring_atomic_$rep.json 2> $OUT/ring_atomic_$rep.err
done
python - <<PY
import json,glob,os
out=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r6_call13"
for f in sorted(glob.glob(out+"/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        r=d.get("roofline",{})
        print(os.path.basename(f), d["ms_per_step"], "ms | dominant", r.get("kernel"), r.get("avg_ms"), "ms frac", r.get("frac"))
    except Exception as e:
        print(os.path.basename(f), "ERR", e, open(f.replace(".json",".err")).read()[-600:])
PY
