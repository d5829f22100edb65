#!/bin/bash
# round 6, call 6: which stream holds the memory-only loop back (ablation bits on top of the no-arithmetic variant)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6_call6; rm -rf $OUT; mkdir -p $OUT
python scripts/micro/dump_topo.py water /tmp/topo_water.bin > $OUT/dump.log 2>&1
L=scripts/micro/lab
{
for v in nopf mo_nopf mo_a1 mo_a2 mo_a3 mo_a4 mo_a7 mo_a8 mo_a11 mo_a12 mo_a15; do
  printf "%-10s " $v; timeout 120 $L/$v.out /tmp/topo_water.bin 20 64 0 4 2>/dev/null
done
} > $OUT/lab_times.txt 2>&1
cat $OUT/lab_times.txt
