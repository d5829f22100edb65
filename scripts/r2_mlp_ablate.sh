# radial-MLP ablations (timing probes, wrong results).  forward NQA_MLP_DBG bits: 1 no stores, 2 no staging, 4 no barrier,
# 32 no LDS fragment reads, 64 NW=8.  backward NQA_MLP_DBG_BWD bits: 1 no g_w loads, 2 no weight staging, 4 no barrier,
# 16 no LDS weight-fragment reads after the first chunk (middle layer 303-308 -> 264-284 us)
cd $GRAFT_REPO_ROOT
for d in 0 1 2 3 7; do
  echo "== NQA_MLP_DBG_BWD=$d"; E=200279 NQA_MLP_DBG_BWD=$d timeout 120 python scripts/bench_mlp.py 2>&1 | grep "H=128" | sed 's/| rocBLAS.*//'
done
