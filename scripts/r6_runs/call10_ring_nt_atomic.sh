#!/bin/bash
# round 6, call 10: ring kernel with non-temporal weight copies / atomic grad_x of the other node
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6_call10; rm -rf $OUT; mkdir -p $OUT
python scripts/micro/dump_topo.py water /tmp/topo_water.bin > $OUT/dump.log 2>&1
L=scripts/micro/lab
{
for v in ring ring_ntw ring_ntwxg; do for w in 4 2 1; do printf "%-14s " $v; timeout 120 $L/$v.out /tmp/topo_water.bin 20 64 0 $w 2>/dev/null; done; done
for v in ring_gxat ring_gxat_ntw; do for w in 4 2 1; do printf "%-14s " $v; timeout 120 $L/$v.out /tmp/topo_water.bin 20 64 0 $w 0 2.25 0 0 1 2>/dev/null; done; done
echo "--- morton relabel 4.5"
for v in ring ring_ntw; do for w in 2 1; do printf "%-14s " $v; timeout 120 $L/$v.out /tmp/topo_water.bin 20 64 1 $w 0 4.5 2>/dev/null; done; done
} > $OUT/lab_times.txt 2>&1
cat $OUT/lab_times.txt
