#!/bin/bash
# round 6, call 22: spatial node order inside the model: test, then cfg-3 / cu100k steps with NQA_SPATIAL_ORDER=0/1
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6_call22; rm -rf $OUT; mkdir -p $OUT
python -m pytest tests/test_model_properties_gpu.py tests/test_baseline_size_parity.py -x -q -m gpu -k "spatial or cfg5 or cfg3" 2>&1 | tail -4 > $OUT/tests.txt; cat $OUT/tests.txt
for rep in 1 2; do for v in 0 1; do
  NQA_SPATIAL_ORDER=$v python bench.py --workload cu20k --steps 10 --warmup 3 --no-cpu-baseline --no-pmc --no-other-workloads 2>/dev/null | tail -1 > $OUT/cu20k_sp${v}_$rep.json
  NQA_SPATIAL_ORDER=$v python bench.py --workload cu100k --steps 5 --warmup 2 --no-cpu-baseline --no-pmc --no-other-workloads 2>/dev/null | tail -1 > $OUT/cu100k_sp${v}_$rep.json
done; done
python - <<PY
import json,glob,os
out=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r6_call22"
for f in sorted(glob.glob(out+"/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); k=d["kernels_ms_per_step"]
        print(os.path.basename(f), "%.3f ms" % d["ms_per_step"], {a: round(b,3) for a,b in k.items() if a.startswith("tp")})
    except Exception as e:
        print(os.path.basename(f), "ERR", e)
PY
