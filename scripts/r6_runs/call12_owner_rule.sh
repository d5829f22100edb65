#!/bin/bash
# round 6, call 12: who owns a pair -- balanced parity rule (as built) vs the smaller / larger node index (one-sided halo)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6_call12; rm -rf $OUT; mkdir -p $OUT
python scripts/micro/dump_topo.py water /tmp/topo_water.bin > $OUT/dump.log 2>&1
L=scripts/micro/lab
{
for rule in 0 1 2; do
  for w in 4 2 1; do printf "%-14s rule %d " ring_ntw $rule; timeout 120 $L/ring_ntw.out /tmp/topo_water.bin 20 64 0 $w 0 2.25 0 0 0 $rule 2>&1 | grep -v "^mean"; done
  for w in 2 1; do printf "%-14s rule %d " ring_gxat_ntw $rule; timeout 120 $L/ring_gxat_ntw.out /tmp/topo_water.bin 20 64 0 $w 0 2.25 0 0 1 $rule 2>&1 | grep -v "^mean"; done
  printf "%-14s rule %d " base $rule; timeout 120 $L/base.out /tmp/topo_water.bin 20 64 0 4 0 2.25 0 0 0 $rule 2>&1 | grep -v "^mean"
done
echo "--- morton 4.5 + rule 1"
for w in 2 1; do printf "%-14s rule 1 " ring_ntw; timeout 120 $L/ring_ntw.out /tmp/topo_water.bin 20 64 1 $w 0 4.5 0 0 0 1 2>&1 | grep -v "^mean"; done
} > $OUT/lab_times.txt 2>&1
cat $OUT/lab_times.txt
