#!/bin/bash
# round 6, call 18: forward on the LDS ring: tests of the spec kernels, then the step with NQA_FWD_RING=0/1 (cfg-3, cu20k)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6_call18; rm -rf $OUT; mkdir -p $OUT
python -m pytest tests/test_tp_spec_kernels.py tests/test_tp_scatter_kernel.py -x -q -m gpu 2>&1 | tail -4 > $OUT/tests.txt
cat $OUT/tests.txt
for w in water10k cu20k; do for rep in 1 2; do
  for v in 0 1; do
    NQA_FWD_RING=$v python bench.py --workload $w --steps 30 --warmup 5 --no-cpu-baseline --no-pmc --no-other-workloads 2>/dev/null | tail -1 > $OUT/${w}_fring${v}_$rep.json
  done
done; done
python - <<PY
import json,glob,os
out=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r6_call18"
for f in sorted(glob.glob(out+"/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); k=d["kernels_ms_per_step"]
        print(os.path.basename(f), "%.3f ms" % d["ms_per_step"], "tp_fwd %.3f" % k.get("tp_fwd",0))
    except Exception as e:
        print(os.path.basename(f), "ERR", e)
PY
