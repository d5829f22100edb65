#!/bin/bash
# SQ counter passes over the radial-MLP micro-benchmark (diagnostic)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_mlp; rm -rf $OUT; mkdir -p $OUT
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/$tag -o p -- python $GRAFT_REPO_ROOT/scripts/bench_mlp.py > /dev/null 2> $OUT/$tag.err
done
python - <<PY
import csv,glob,os,collections
out=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/pmc_mlp"
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(out+"/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if "bf16x6" not in r["Kernel_Name"]: continue
        key=("fwd" if "fwd" in r["Kernel_Name"] else "bwd", r.get("Grid_Size", r.get("Grid_Size_X","?")))
        agg[key][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[key][r["Counter_Name"]]+=1
for k,v in sorted(agg.items()):
    print(k, {c:round(x/cnt[k][c]) for c,x in v.items()})
PY
