#!/bin/bash
# round 6, call 19: cfg-3 middle layer as TWO wavefronts per (node): split by input block on the ring (fewer nodes in flight per XCD)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6_call19; rm -rf $OUT; mkdir -p $OUT
python scripts/micro/dump_topo.py water /tmp/topo_water.bin > $OUT/dump.log 2>&1
L=scripts/micro/lab
{
for w in 4 1; do printf "%-10s gxat 1 " ring; timeout 120 $L/ring.out /tmp/topo_water.bin 20 64 0 $w 0 2.25 0 0 1 2>&1 | grep -v "^mean"; done
for v in ${SR_VARIANTS:-l2sr}; do for gx in 0 1; do
  printf "%-10s gxat %d " $v $gx; timeout 120 $L/$v.out /tmp/topo_water.bin 20 64 0 4 0 2.25 0 0 $gx 2>&1 | grep -v "^mean"
done; done
} > $OUT/lab_times.txt 2>&1
cat $OUT/lab_times.txt
