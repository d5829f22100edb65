# node_linear ablation (timing probes): NQA_NODE_DBG bits 1 no global loads, 2 one MFMA batch only, 4 no output stores,
# 8 no result round trip through LDS, 16 no barriers, 32 no LDS staging stores
cd $GRAFT_REPO_ROOT
for d in 0 1 2 4 6 7 63; do
  echo "== NQA_NODE_DBG=$d"; NQA_NODE_DBG=$d timeout 120 python scripts/bench_linear.py 2>&1 | grep "Z=10125"
done
