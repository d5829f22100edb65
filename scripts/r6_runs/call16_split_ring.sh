#!/bin/bash
# round 6, call 16: split pair kernel (l3n_mid, 128 channels, cu20k lists) -- plain loop vs LDS ring, rows vs accumulator
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6_call16; rm -rf $OUT; mkdir -p $OUT
python scripts/micro/dump_topo.py cu20k /tmp/topo_cu20k.bin > $OUT/dump.log 2>&1
L=scripts/micro/lab
{
for v in cu ${SR_VARIANTS:-cusr}; do for gx in 0 1; do
  [ $v = cu ] && [ $gx = 1 ] && continue
  printf "%-10s gxat %d " $v $gx; timeout 300 $L/$v.out /tmp/topo_cu20k.bin 5 128 0 4 0 2.25 0 0 $gx 2>&1 | grep -v "^mean"
done; done
} > $OUT/lab_times.txt 2>&1
cat $OUT/lab_times.txt
