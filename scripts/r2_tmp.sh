cd $GRAFT_REPO_ROOT
WORKLOADS="water10k cu20k" bash scripts/r2_quick_bench.sh
WORKLOADS="water10k" NQA_NO_OVERLAP=1 bash scripts/r2_quick_bench.sh
timeout 900 python -m pytest tests/test_tp_spec_kernels.py tests/test_tp_scatter_kernel.py tests/test_edge_pairs.py -x -q -m gpu 2>&1 | tail -2
