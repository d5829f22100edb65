# timing probes of the fused tensor-product backward (gen_spec.py NQA_GEN_PROBE bits: 1 no grad_w stores, 2 no weight
# loads, 4 no per-edge grad_x rows, 8 no x gather; wrong results).  The variants are built beforehand:
#   for P in 1 2 ...; do NQA_GEN_PROBE=$P python -m nequip_amd.csrc.build; cp nequip_amd/csrc/libnequip_amd.so scripts/micro/libprobe_$P.so.out; done
cd $GRAFT_REPO_ROOT
cp nequip_amd/csrc/libnequip_amd.so /tmp/lib_keep.so
for P in ${PROBES:-0 1 2 3 4 7 15 0}; do
  cp scripts/micro/libprobe_$P.so.out nequip_amd/csrc/libnequip_amd.so
  echo -n "probe $P: "; bash scripts/r2_quick_bench.sh | python -c "
import sys,re
l=sys.stdin.read()
m=re.search(r'ms/step ([0-9.]+)',l); f=re.search(r\"'tp_bwd_fused': ([0-9.]+)\",l); b=re.search(r\"'radial_mlp_bwd': ([0-9.]+)\",l)
print('step', m.group(1), 'tp_bwd_fused', f.group(1), 'radial_mlp_bwd', b.group(1))"
done
cp /tmp/lib_keep.so nequip_amd/csrc/libnequip_amd.so
