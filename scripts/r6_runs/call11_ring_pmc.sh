#!/bin/bash
# round 6, call 11: L2 traffic of the ring kernel by node order (raster as built / Morton) and with atomics
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6_call11; rm -rf $OUT; mkdir -p $OUT
python scripts/micro/dump_topo.py water /tmp/topo_water.bin > $OUT/dump.log 2>&1
L=$GRAFT_REPO_ROOT/scripts/micro/lab
cd /tmp
run() {  # tag, binary, args...
  tag=$1; shift
  for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum"; do
    t=$(echo $set | tr ' ' '_' | cut -c1-40)
    timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pmc/$tag/$t -o p -- "$@" > /dev/null 2> $OUT/err_${tag}_$t.txt
  done
}
run ring_raster $L/ring_ntw.out /tmp/topo_water.bin 3 64 0 1
run ring_morton45 $L/ring_ntw.out /tmp/topo_water.bin 3 64 1 1 0 4.5
run ring_morton9 $L/ring_ntw.out /tmp/topo_water.bin 3 64 1 1 0 9.0
run ring_wpn4 $L/ring_ntw.out /tmp/topo_water.bin 3 64 0 4
run gxat_raster $L/ring_gxat_ntw.out /tmp/topo_water.bin 3 64 0 1 0 2.25 0 0 1
python - <<PY
import csv,glob,os,collections
out=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r6_call11"
with open(out+"/ring_pmc.txt","w") as fo:
    for tag in sorted(os.listdir(out+"/pmc")):
        agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(lambda: collections.defaultdict(int))
        for f in glob.glob(out+"/pmc/"+tag+"/**/*counter_collection.csv",recursive=True):
            for r in csv.DictReader(open(f)):
                n=r["Kernel_Name"]
                key="pair" if "bwd_pair" in n else ("sum" if "gx_rows_sum" in n else None)
                if key is None: continue
                agg[key][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[key][r["Counter_Name"]]+=1
        for k,v in sorted(agg.items()):
            fo.write(f"{tag:14s} {k:5s} " + "  ".join(f"{c}={x/cnt[k][c]:.0f}" for c,x in sorted(v.items())) + "\n")
print(open(out+"/ring_pmc.txt").read())
PY
rm -rf $OUT/pmc
