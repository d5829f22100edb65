cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_radial_mlp.py tests/test_reference_golden.py tests/test_node_kernels.py -x -q -m gpu 2>&1 | tail -3
for b in 1 0; do echo "balanced=$b"; E=200279 NQA_MLP_FWD_BALANCED=$b python scripts/bench_mlp.py 2>&1 | sed 's/| rocBLAS.*//'; done
bash scripts/r2_quick_bench.sh
