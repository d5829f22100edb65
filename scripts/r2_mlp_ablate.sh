# radial-MLP forward ablation (timing probes; bits: 1 no stores, 2 no staging, 4 no barrier, 32 no LDS fragment reads, 64 NW=8)
cd $GRAFT_REPO_ROOT
for d in 0 1 3 7 39 64 65; do
  echo "== NQA_MLP_DBG=$d"; E=200279 NQA_MLP_DBG=$d timeout 120 python scripts/bench_mlp.py 2>&1 | grep "H=128 W=704" | sed 's/| rocBLAS.*//'
done
