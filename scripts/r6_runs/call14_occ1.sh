#!/bin/bash
# round 6, call 14: ring kernel at one workgroup per CU (extra LDS): fewer nodes in flight per XCD -> L2 hit rate of the gathers
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6_call14; rm -rf $OUT; mkdir -p $OUT
python scripts/micro/dump_topo.py water /tmp/topo_water.bin > $OUT/dump.log 2>&1
L=scripts/micro/lab
{
for x in 0 40; do for w in 4 2 1; do printf "%-14s lds+%2d " ring $x; timeout 120 $L/ring.out /tmp/topo_water.bin 20 64 0 $w 0 2.25 $x 0 1 2>&1 | grep -v "^mean"; done; done
for x in 0 40; do for w in 4 1; do printf "%-14s lds+%2d " ring $x; timeout 120 $L/ring.out /tmp/topo_water.bin 20 64 0 $w 0 2.25 $x 0 0 2>&1 | grep -v "^mean"; done; done
} > $OUT/lab_times.txt 2>&1
cat $OUT/lab_times.txt
