#!/bin/bash
# round 4, GPU call 9: the rest of the GPU suite (from test_eval_mode_weights on) after the cache fix + the no-masking fast
# paths of the node kernels, then the bench
mkdir -p gpurun_out/r4i
python -m pytest -q -m gpu tests/test_eval_mode_weights.py tests/test_full_size_properties.py tests/test_ghost_exchange.py tests/test_golden_gpu.py tests/test_mlp_training_fn.py tests/test_model_parity.py tests/test_model_properties_gpu.py tests/test_neighbor_list.py tests/test_node_fused.py tests/test_node_kernels.py tests/test_norm_activation.py tests/test_presets.py tests/test_radial_mlp.py tests/test_reference_golden.py tests/test_topology_cache.py tests/test_tp_scatter_kernel.py tests/test_tp_scatter_ops.py tests/test_tp_spec_kernels.py tests/test_traceable_model.py tests/test_training_step.py tests/test_wgrad.py tests/test_edge_embed.py tests/test_edge_pairs.py tests/test_edge_vectors_dtypes.py tests/test_ddp_rccl.py tests/test_ddp_model_shared_device.py tests/test_bench_contract.py > gpurun_out/r4i/tests.log 2>&1
echo "rc=$?" >> gpurun_out/r4i/tests.log
tail -8 gpurun_out/r4i/tests.log
for i in 1 2; do
  python bench.py --no-pmc --no-cpu-baseline > gpurun_out/r4i/bench_$i.json 2> gpurun_out/r4i/bench_$i.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4i/bench_*.json')):
    d=json.loads([l for l in open(f) if l.startswith('{')][0])
    print(f, round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['kernels_ms_per_step'].items()})
PY
