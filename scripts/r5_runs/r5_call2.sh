#!/bin/bash
# round 5, call 2: SQ counters of the f16x3 radial-MLP kernels alone (where do the wave cycles go?)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5c2; rm -rf $OUT; mkdir -p $OUT
python $GRAFT_REPO_ROOT/scripts/bench_mlp.py > $OUT/mlp_alone.log 2>&1
cat $OUT/mlp_alone.log | grep "H="
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA" "GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM SQ_WAIT_INST_ANY"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/$tag -o p -- python $GRAFT_REPO_ROOT/scripts/bench_mlp.py > /dev/null 2> $OUT/$tag.err
done
python - <<PY
import csv,glob,os,collections
out=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r5c2"
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(out+"/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        kn=r["Kernel_Name"]
        if "radial_mlp" not in kn or "split_w1" in kn: continue
        key=("fwd" if "fwd" in kn else "bwd", r.get("Grid_Size", r.get("Grid_Size_X","?")))
        agg[key][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[key][r["Counter_Name"]]+=1
with open(out+"/summary.txt","w") as fo:
    for k,v in sorted(agg.items()):
        line=str(k)+" "+str({c:round(x/cnt[k][c]) for c,x in v.items()})
        print(line); fo.write(line+"\n")
PY

