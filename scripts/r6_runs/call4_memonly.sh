#!/bin/bash
# round 6, call 4: the memory streams alone (no arithmetic) with a second pair in flight / more wavefronts per SIMD
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6_call4; rm -rf $OUT; mkdir -p $OUT
python scripts/micro/dump_topo.py water /tmp/topo_water.bin > $OUT/dump.log 2>&1
L=scripts/micro/lab
{
for v in base memonly mo_nopf mo_pipe mo_occ4 mo_occ8 mo_pipe_occ4; do
  printf "%-12s " $v; timeout 120 $L/$v.out /tmp/topo_water.bin 20 64 0 4
done
echo "--- wpn 1"
for v in memonly mo_pipe mo_occ4; do printf "%-12s " $v; timeout 120 $L/$v.out /tmp/topo_water.bin 20 64 0 1; done
} > $OUT/lab_times.txt 2>&1
cat $OUT/lab_times.txt
