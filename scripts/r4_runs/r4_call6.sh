#!/bin/bash
# round 4, GPU call 6: the other workloads on the round-4 build
mkdir -p gpurun_out/r4f
for w in si1k aspirin5 cu20k cu100k train256 water10k_S water10k_M water10k_L water10k_XL; do
  timeout 400 python bench.py --workload $w --no-cpu-baseline --no-pmc 2> gpurun_out/r4f/$w.err >> gpurun_out/r4f/r4_other_workloads.jsonl
done
python - <<'PY'
import json
for l in open('gpurun_out/r4f/r4_other_workloads.jsonl'):
    if not l.startswith('{'): continue
    d=json.loads(l)
    print(d['config']['workload'][:40], round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['kernels_ms_per_step'].items()})
PY
