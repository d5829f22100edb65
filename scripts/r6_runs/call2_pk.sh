#!/bin/bash
# round 6, call 2: packed-fp32 pair kernel against the scalar one (lab), packed FMA rate micro-benchmark
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6_call2; rm -rf $OUT; mkdir -p $OUT
python scripts/micro/dump_topo.py water /tmp/topo_water.bin > $OUT/dump.log 2>&1
L=scripts/micro/lab
{
$L/pk_rate.out
for v in base pk pkpf pk3 pk4 pk8 pk15 pk31 spf2 base; do
  printf "%-10s " $v; timeout 120 $L/$v.out /tmp/topo_water.bin 20 64 0 4
done
echo "--- wpn 1"
for v in pk; do printf "%-10s " $v; timeout 120 $L/$v.out /tmp/topo_water.bin 20 64 0 1; done
} > $OUT/lab_times.txt 2>&1
cat $OUT/lab_times.txt
