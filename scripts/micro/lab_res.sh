#!/bin/bash
# registers / spills / code size of the pair kernel in every scripts/micro/lab/*.out (no GPU)
cd "$(dirname "$0")/lab" || exit 1
T=$(mktemp -d)
for f in ${@:-*.out}; do
  cp $f $T/x.out; (cd $T && /opt/rocm/lib/llvm/bin/llvm-objdump --offloading x.out >/dev/null 2>&1)
  CO=$(ls $T/x.out.*amdgcn* 2>/dev/null | head -1)
  [ -z "$CO" ] && continue
  /opt/rocm/lib/llvm/bin/llvm-readelf --notes $CO | awk -v f=$f '/\.name:/{n=$2} /\.vgpr_count:/{v=$2} /\.vgpr_spill_count:/{s=$2} /\.sgpr_count:/{sg=$2} /\.private_segment_fixed_size:/{p=$2} /\.group_segment_fixed_size:/{l=$2} /\.wavefront_size:/{ if (n ~ /pair/ && n ~ /Li4E|split/) printf "%-22s vgpr %3d spill %3d sgpr %3d scratch %4d lds %5d  %s\n", f, v, s, sg, p, l, substr(n, 20, 40)}'
  rm -f $T/x.out*
done
