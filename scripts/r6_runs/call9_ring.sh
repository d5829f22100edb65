#!/bin/bash
# round 6, call 9: first run of the LDS-ring pair kernel (lab harness; checksums against the register kernel)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6_call9; rm -rf $OUT; mkdir -p $OUT
python scripts/micro/dump_topo.py water /tmp/topo_water.bin > $OUT/dump.log 2>&1
L=scripts/micro/lab
{
printf "%-10s " base; timeout 120 $L/base.out /tmp/topo_water.bin 20 64 0 4 2>/dev/null
for v in ${RING_VARIANTS:-ring}; do for w in 4 2 1; do printf "%-10s " $v; timeout 120 $L/$v.out /tmp/topo_water.bin 20 64 0 $w 2>/dev/null; done; done
} > $OUT/lab_times.txt 2>&1
cat $OUT/lab_times.txt
