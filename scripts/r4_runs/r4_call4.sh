#!/bin/bash
# round 4, GPU call 4: pipelined stage loop of the node kernels (one memory wait per stage), rocPRIM sorts, per-edge-type
# cutoffs / trained Bessel roots -- parity first, then same-box A/B
mkdir -p gpurun_out/r4d
python -m pytest -x -q -m gpu tests/test_node_fused.py tests/test_node_kernels.py tests/test_edge_pairs.py tests/test_tp_scatter_kernel.py tests/test_neighbor_list.py tests/test_topology_cache.py tests/test_reference_golden.py tests/test_model_parity.py tests/test_full_size_properties.py > gpurun_out/r4d/tests.log 2>&1
echo "rc=$?" >> gpurun_out/r4d/tests.log
tail -12 gpurun_out/r4d/tests.log
for i in 1 2; do
  python bench.py --no-pmc --no-cpu-baseline > gpurun_out/r4d/bench_pipe_$i.json 2> gpurun_out/r4d/bench_pipe_$i.err
  NQA_NODE_PIPE=0 python bench.py --no-pmc --no-cpu-baseline > gpurun_out/r4d/bench_nopipe_$i.json 2> gpurun_out/r4d/bench_nopipe_$i.err
done
python bench.py --no-pmc --no-cpu-baseline --workload cu20k > gpurun_out/r4d/bench_cu20k_pipe.json 2> gpurun_out/r4d/bench_cu20k_pipe.err
NQA_NODE_PIPE=0 python bench.py --no-pmc --no-cpu-baseline --workload cu20k > gpurun_out/r4d/bench_cu20k_nopipe.json 2> gpurun_out/r4d/bench_cu20k_nopipe.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4d/bench_*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][0])
        print(f, round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['kernels_ms_per_step'].items() if k.startswith(('node','gate','energy'))})
    except Exception as e:
        print(f, 'ERR', e)
PY
