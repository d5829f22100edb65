#!/bin/bash
mkdir -p gpurun_out/r4y
python -m pytest -x -q -m gpu tests/test_edge_pairs.py tests/test_cpp_torch_ops.py tests/test_model_parity.py tests/test_topology_cache.py tests/test_ghost_exchange.py tests/test_traceable_model.py tests/test_neighbor_list.py > gpurun_out/r4y/tests.log 2>&1
echo "rc=$?" >> gpurun_out/r4y/tests.log
tail -6 gpurun_out/r4y/tests.log | cut -c 1-1200
python scripts/bench_topo.py 2>&1 | tail -7
python bench.py --no-pmc --no-cpu-baseline > gpurun_out/r4y/bench.json 2> gpurun_out/r4y/bench.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r4y/bench.json') if l.startswith('{')][0])
print('bench', round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['kernels_ms_per_step'].items() if k.startswith('tp')})
PY
python scripts/bench_md.py 2>&1 | tail -1
timeout 600 python scripts/bench_deployed.py --no-aoti 2>/dev/null | grep '"form"' | cut -c 1-120
