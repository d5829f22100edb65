#!/bin/bash
# round 6, call 5: pair rows numbered in owner-slot order (sequential w / grad_w streams) and the other node's grad_x by atomics
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6_call5; rm -rf $OUT; mkdir -p $OUT
python scripts/micro/dump_topo.py water /tmp/topo_water.bin > $OUT/dump.log 2>&1
L=scripts/micro/lab
{
for v in base nopf memonly mo_nopf mo_pipe; do
  for prid in 0 1; do printf "%-10s prid %d " $v $prid; timeout 120 $L/$v.out /tmp/topo_water.bin 20 64 0 4 0 2.25 0 $prid 0; done
done
for v in gxat gxat_nopf; do
  for prid in 0 1; do printf "%-10s prid %d " $v $prid; timeout 120 $L/$v.out /tmp/topo_water.bin 20 64 0 4 0 2.25 0 $prid 1; done
done
echo "--- wpn 1"
for v in nopf; do for prid in 0 1; do printf "%-10s prid %d " $v $prid; timeout 120 $L/$v.out /tmp/topo_water.bin 20 64 0 1 0 2.25 0 $prid 0; done; done
for v in gxat_nopf; do for prid in 0 1; do printf "%-10s prid %d " $v $prid; timeout 120 $L/$v.out /tmp/topo_water.bin 20 64 0 1 0 2.25 0 $prid 1; done; done
} > $OUT/lab_times.txt 2>&1
cat $OUT/lab_times.txt
