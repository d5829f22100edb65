#!/bin/bash
# round 6, call 21: the small radial backward (LDS-resident fragments) for the exposed layer-0 launch, re-measured on this round's step
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6_call21; rm -rf $OUT; mkdir -p $OUT
B="python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-pmc --no-other-workloads"
for rep in 1 2 3; do for v in 0 1 2; do
  NQA_MLP_BWD_SMALL=$v $B 2>/dev/null | tail -1 > $OUT/small${v}_$rep.json
done; done
python - <<PY
import json,glob,os
out=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r6_call21"
for f in sorted(glob.glob(out+"/*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); k=d["kernels_ms_per_step"]
    print(os.path.basename(f), "%.4f ms" % d["ms_per_step"], "radial_mlp_bwd %.3f" % k.get("radial_mlp_bwd",0))
PY
