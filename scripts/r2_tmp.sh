cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_node_chain.py -x -q -m gpu 2>&1 | tail -2
for g in 16 40; do echo "NQA_CHAIN_G=$g"; NQA_CHAIN_G=$g bash scripts/r2_quick_bench.sh; done
echo "NQA_NO_CHAIN=1"; NQA_NO_CHAIN=1 bash scripts/r2_quick_bench.sh
