#!/bin/bash
# SQ counter passes over the cfg-3 bench (eager, 2 steps): per-kernel averages for the tensor-product kernels (diagnostic)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_tp; rm -rf $OUT; mkdir -p $OUT
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/$tag -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --kernel-steps 0 > /dev/null 2> $OUT/$tag.err
done
python - <<PY
import csv,glob,os,collections,re
out=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/pmc_tp"
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(out+"/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        n=r["Kernel_Name"]
        if "SpecArgs" not in n: continue
        m=re.search(r"(fwd_kernel|bwd_edge_kernel<float, 4, (?:true|false), (?:true|false), (?:true|false)>|bwd_x_kernel|gx_rows_sum_kernel)", n)
        key=(m.group(1) if m else n[:40], r.get("Grid_Size", r.get("Grid_Size_X","?")))
        agg[key][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[key][r["Counter_Name"]]+=1
for k,v in sorted(agg.items()):
    print(k, {c:round(x/cnt[k][c]) for c,x in v.items()})
PY
