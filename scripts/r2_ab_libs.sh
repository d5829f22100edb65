# same-box A/B of prebuilt library variants scripts/micro/libprobe_<P>.so.out: PROBES="0 1 0 1" WORKLOADS="water10k cu20k"
cd $GRAFT_REPO_ROOT
cp nequip_amd/csrc/libnequip_amd.so /tmp/lib_keep.so
for P in ${PROBES:-0 1 0 1}; do
  cp scripts/micro/libprobe_$P.so.out nequip_amd/csrc/libnequip_amd.so
  echo "variant $P:"; bash scripts/r2_quick_bench.sh
done
cp /tmp/lib_keep.so nequip_amd/csrc/libnequip_amd.so
