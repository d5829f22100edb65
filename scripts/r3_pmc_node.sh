#!/bin/bash
# SQ / GRBM counter passes over scripts/r3_node_bench.py (one shape): per-dispatch averages of the node kernel.
# usage: bash scripts/r3_pmc_node.sh <only-filter> [extra args of r3_node_bench.py]
ONLY=${1:-layer1.linear_2}; shift
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_node; rm -rf $OUT; mkdir -p $OUT
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/$tag -o p -- python $GRAFT_REPO_ROOT/scripts/r3_node_bench.py --only $ONLY --reps 20 "$@" > /dev/null 2> $OUT/$tag.err
done
python - <<PY
import csv,glob,os,collections
out=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/pmc_node"
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(out+"/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        n=r["Kernel_Name"]
        if "node_linear" not in n: continue
        key=(n[:60], r.get("Grid_Size","?"))
        agg[key][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[key][r["Counter_Name"]]+=1
for k,v in sorted(agg.items()):
    print(k)
    for c,x in sorted(v.items()): print("   %-32s %14.0f  (n=%d)" % (c, x/cnt[k][c], cnt[k][c]))
PY
