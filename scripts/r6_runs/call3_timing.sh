#!/bin/bash
# round 6, call 3: where the cycles of one pair go (s_memtime), the memory streams without the arithmetic, occupancy limits
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6_call3; rm -rf $OUT; mkdir -p $OUT
python scripts/micro/dump_topo.py water /tmp/topo_water.bin > $OUT/dump.log 2>&1
L=scripts/micro/lab
{
for v in base tm tm31 tmmem memonly; do
  printf "%-10s " $v; timeout 120 $L/$v.out /tmp/topo_water.bin 20 64 0 4
done
echo "--- wpn 1"
for v in tm tmmem; do printf "%-10s " $v; timeout 120 $L/$v.out /tmp/topo_water.bin 20 64 0 1; done
echo "--- occupancy: extra LDS per workgroup 70 KiB (2 workgroups per CU = 2 wavefronts per SIMD), 150 KiB (1 per CU)"
for v in base memonly; do for k in 0 40 70 150; do printf "%-10s lds+%3d " $v $k; timeout 120 $L/$v.out /tmp/topo_water.bin 20 64 0 4 0 2.25 $k; done; done
} > $OUT/lab_times.txt 2>&1
cat $OUT/lab_times.txt
