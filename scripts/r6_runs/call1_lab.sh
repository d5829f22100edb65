#!/bin/bash
# round 6, call 1: ablations, generator variants and SQ / TCC / TCP counters of the pair-centric backward (cfg-3 middle layer)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6_call1; rm -rf $OUT; mkdir -p $OUT
python scripts/micro/dump_topo.py water /tmp/topo_water.bin > $OUT/dump.log 2>&1
python scripts/micro/dump_topo.py cu20k /tmp/topo_cu20k.bin >> $OUT/dump.log 2>&1
L=scripts/micro/lab
{
for v in base abl1 abl2 abl3 abl4 abl8 abl9 abl11 abl12 abl15 abl16 abl31 nopf nont ntl nohoist occ3 occ4 nohoist3 nohoist4 pipe pipenh; do
  printf "%-10s " $v; timeout 120 $L/$v.out /tmp/topo_water.bin 20 64 0 4
done
for v in split2 split3; do printf "%-10s " $v; timeout 120 $L/$v.out /tmp/topo_water.bin 20 64 0 4; done
echo "--- wpn 1"
for v in base nopf split2; do printf "%-10s " $v; timeout 120 $L/$v.out /tmp/topo_water.bin 20 64 0 1; done
echo "--- relabel morton (2.25 / 4.5 / 9.0) / random"
printf "%-10s " base; timeout 120 $L/base.out /tmp/topo_water.bin 20 64 1 4 0 2.25
printf "%-10s " base; timeout 120 $L/base.out /tmp/topo_water.bin 20 64 1 4 0 4.5
printf "%-10s " base; timeout 120 $L/base.out /tmp/topo_water.bin 20 64 1 4 0 9.0
printf "%-10s " base; timeout 120 $L/base.out /tmp/topo_water.bin 20 64 2 4
echo "--- modes (pair only / sum only)"
printf "%-10s " base; timeout 120 $L/base.out /tmp/topo_water.bin 20 64 0 4 1
printf "%-10s " base; timeout 120 $L/base.out /tmp/topo_water.bin 20 64 0 4 2
echo "--- cu20k l3n_mid mul 128"
printf "%-10s " cu; timeout 300 $L/cu.out /tmp/topo_cu20k.bin 5 128 0 4
printf "%-10s " cu; timeout 300 $L/cu.out /tmp/topo_cu20k.bin 5 128 1 4 0 4.5
} > $OUT/lab_times.txt 2>&1
cat $OUT/lab_times.txt
# counters of the baseline kernel alone
cd /tmp
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum TCC_EA_WRREQ_sum TCC_EA_WRREQ_64B_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum" "TA_BUSY_avr TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TCC_TAG_STALL_sum TCC_BUSY_sum TCC_EA_WR_UNCACHED_32B_sum" "FETCH_SIZE" "WRITE_SIZE" "TCC_EA_RD_UNCACHED_32B_sum TCC_NORMAL_WRITEBACK_sum TCC_NORMAL_EVICT_sum" "SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_INSTS_FLAT"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-48)
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pmc/$tag -o p -- $GRAFT_REPO_ROOT/$L/base.out /tmp/topo_water.bin 5 64 0 4 > /dev/null 2> $OUT/pmc_$tag.err
done
python - <<PY
import csv,glob,os,collections
out=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r6_call1"
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(out+"/pmc/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        n=r["Kernel_Name"]
        key="pair" if "bwd_pair" in n else ("sum" if "gx_rows_sum" in n else None)
        if key is None: continue
        agg[key][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[key][r["Counter_Name"]]+=1
with open(out+"/pair_counters.txt","w") as fo:
    for k,v in sorted(agg.items()):
        for c,x in sorted(v.items()):
            fo.write(f"{k:5s} {c:40s} {x/cnt[k][c]:18.1f}  (n={cnt[k][c]})\n")
print(open(out+"/pair_counters.txt").read())
PY
rm -rf $OUT/pmc
tail -3 $OUT/pmc_*.err | head -80
