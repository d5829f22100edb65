#!/bin/bash
# round 6, call 20: forward of the l_max = 3 structures split by input block: tests, cu20k / cu100k steps with NQA_FWD_SPLIT=0/1
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6_call20; rm -rf $OUT; mkdir -p $OUT
python -m pytest tests/test_tp_spec_kernels.py tests/test_tp_scatter_kernel.py tests/test_baseline_size_parity.py -q -m gpu -k "l3 or cfg5 or one_wavefront or large_degree" 2>&1 | tail -4 > $OUT/tests.txt
cat $OUT/tests.txt
for w in cu20k cu100k; do for rep in 1 2; do
  for v in 0 1; do
    NQA_FWD_SPLIT=$v python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-pmc --no-other-workloads 2>/dev/null | tail -1 > $OUT/${w}_fsplit${v}_$rep.json
  done
done; done
python - <<PY
import json,glob,os
out=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r6_call20"
for f in sorted(glob.glob(out+"/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); k=d["kernels_ms_per_step"]
        print(os.path.basename(f), "%.2f ms" % d["ms_per_step"], "tp_fwd %.2f" % k.get("tp_fwd",0))
    except Exception as e:
        print(os.path.basename(f), "ERR", e)
PY
