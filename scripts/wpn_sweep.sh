#!/bin/bash
for v in 4 1 4 1; do
  NQA_SPEC_WPN=$v python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; o=r['other_kernels']; print('wpn$v', round(d['ms_per_step'],3), r['kernel'], round(r['avg_launch_ms'],4), 'fwd', round(o['tp_fwd']['avg_launch_ms'],4), 'bwd_edge', round(o['tp_bwd_edge']['avg_launch_ms'],4), 'bwd_x', round(o['tp_bwd_x']['avg_launch_ms'],4))"
done
