#!/bin/bash
mkdir -p gpurun_out/r5c38
for rep in 1 2; do
for v in 0 1; do
  echo "NQA_PAIR_ON_MAIN=$v" >> gpurun_out/r5c38/ab.log
  NQA_PAIR_ON_MAIN=$v timeout 300 python scripts/bench_md.py 2>&1 | grep graphed >> gpurun_out/r5c38/ab.log
done; done
cat gpurun_out/r5c38/ab.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc > gpurun_out/r5c38/bench.json 2> gpurun_out/r5c38/bench.err
python - <<'PY'
import json
r = json.loads(open("gpurun_out/r5c38/bench.json").read().strip().splitlines()[-1])
print("ms_per_step", r["ms_per_step"], "md_step", r["config"]["md_step"])
PY
tail -3 gpurun_out/r5c38/bench.err
