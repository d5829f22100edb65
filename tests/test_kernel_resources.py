"""Register / LDS / spill budget of the kernels on the benchmarked path, read from the built code objects (no GPU).

DESIGN.md argues from these figures (wavefronts per SIMD, workgroups per CU, "no spills"); a change that silently pushes a
headline kernel over a boundary -- a fourth wavefront lost to two more registers, a third workgroup to 1 KiB of LDS, a
spill in a hot loop -- shows up here before it shows up as an unexplained slowdown on the GPU.  The numbers are upper
bounds with the current values in the comments (`python scripts/kernel_resources.py <object>` prints the table)."""
import glob
import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import kernel_resources as kr  # noqa: E402


def _objects(pattern):
    objs = glob.glob(os.path.join(kr.BUILD, pattern))
    if not objs or not os.path.exists(os.path.join(kr.LLVM, "llvm-readelf")):
        pytest.skip("build objects / ROCm LLVM tools not present (run python -m nequip_amd.csrc.build)")
    return objs


def _find(kernels, *needles):
    hits = {n: r for n, r in kernels.items() if all(s in n for s in needles)}
    assert len(hits) == 1, (needles, list(hits))
    return next(iter(hits.values()))


def test_radial_mlp_kernels_fit_their_occupancy_targets():
    ks = kr.kernels_of(_objects("radial_mlp.o")[0])
    # inference defaults (two-plane fp16 split): two workgroups of four wavefronts per CU, nothing spilled
    fwd = _find(ks, "radial_mlp_fwd_split_bal_kernel<128, true, false>")         # 224 VGPRs, 51 200 B
    bwd = _find(ks, "radial_mlp_bwd_split_kernel<128, 0, false, true, false>")   # 188 VGPRs, 78 848 B
    for r in (fwd, bwd):
        assert r["vgpr_spill"] == 0 and r["scratch"] == 0
        assert kr.waves_per_simd(r["vgpr"]) >= 2 and 2 * r["lds"] <= 160 * 1024
    assert fwd["vgpr"] <= 232 and bwd["vgpr"] <= 200
    # the fp16 split must not cost registers against the bf16 split it replaces
    assert fwd["vgpr"] <= _find(ks, "radial_mlp_fwd_split_bal_kernel<128, false, false>")["vgpr"]
    assert bwd["vgpr"] <= _find(ks, "radial_mlp_bwd_split_kernel<128, 0, false, false, false>")["vgpr"]
    # training epilogues on the same main loop
    for tm in (1, 2):
        r = _find(ks, f"radial_mlp_bwd_split_kernel<128, {tm}, false, true, false>")  # 192 / 200 VGPRs
        assert r["vgpr_spill"] == 0 and kr.waves_per_simd(r["vgpr"]) >= 2


    # round 4: the last-layer variants of deeper MLPs (pre-activations as input) stay within the same budgets
    for needle in ("radial_mlp_fwd_split_bal_kernel<128, true, true>", "radial_mlp_fwd_split_bal_kernel<64, true, true>",
                   "radial_mlp_bwd_split_kernel<128, 0, false, true, true>", "radial_mlp_bwd_split_kernel<64, 0, false, true, true>",
                   "radial_mlp_bwd_split_kernel<64, 0, true, true, true>"):
        r = _find(ks, needle)
        assert r["vgpr_spill"] == 0 and r["scratch"] == 0 and kr.waves_per_simd(r["vgpr"]) >= 2


def test_node_kernels_keep_three_or_four_wavefronts_per_simd():
    ks = kr.kernels_of(_objects("node_ops.o")[0])
    f16 = _find(ks, "node_linear_wave_bf16_kernel<true>")    # 141 VGPRs, 4 x 8320 B
    bf16 = _find(ks, "node_linear_wave_bf16_kernel<false>")  # 148
    exact = _find(ks, "node_linear_wave_kernel")             # 124
    assert kr.waves_per_simd(f16["vgpr"]) >= 3 and kr.waves_per_simd(bf16["vgpr"]) >= 3 and kr.waves_per_simd(exact["vgpr"]) >= 4
    assert f16["vgpr"] <= bf16["vgpr"]
    for r in (f16, bf16, exact):
        assert r["vgpr_spill"] == 0 and r["scratch"] == 0
    assert 4 * f16["lds"] <= 160 * 1024  # LDS does not cap the occupancy below the register limit
    for name in ("gate_fwd_kernel<float>", "gate_bwd_kernel<float>"):
        assert _find(ks, name)["vgpr_spill"] == 0
    fused = _find(ks, "node_fused_kernel<false>")  # 160 VGPRs, 4 x (8320 + 1536) B: the fused layer-boundary stage
    assert fused["vgpr_spill"] == 0 and fused["scratch"] == 0 and kr.waves_per_simd(fused["vgpr"]) >= 3
    assert 3 * fused["lds"] <= 160 * 1024


def test_cfg3_tensor_product_kernels():
    """The structure of the cfg-3 middle layer (l_max = 2, SO(3) irreps): forward at four wavefronts per SIMD, the
    pair-centric backward that carries `roofline` without spills at two."""
    ks = kr.kernels_of(_objects("tp_spec_l2n_mid_*.o")[0])
    fwd = _find(ks, "fwd_kernel<float, 4, false>")                    # 107 VGPRs
    assert fwd["vgpr_spill"] == 0 and kr.waves_per_simd(fwd["vgpr"]) >= 4
    pair = _find(ks, "bwd_pair_kernel<float, 4, true, true, false>")  # the tp_bwd_fused kernel of the bench line
    assert pair["vgpr_spill"] == 0 and kr.waves_per_simd(pair["vgpr"]) >= 2
    bx = _find(ks, "bwd_x_kernel<float, 4, false>")
    assert bx["vgpr_spill"] == 0 and kr.waves_per_simd(bx["vgpr"]) >= 4
    # round 6: the LDS-ring pair kernel that now carries `roofline` (one wavefront per node, accumulator form): two wavefronts
    # per SIMD, NOTHING spilled (a scratch reload is a vector load: hipcc's own vmcnt wait would drain the ring every pair)
    for inst in ("bwd_pair_ring_kernel<1, true, true>", "bwd_pair_ring_kernel<1, true, false>", "bwd_pair_ring_kernel<4, true, true>",
                 "bwd_pair_ring_kernel<1, false, false>"):
        ring = _find(ks, inst)  # 247 VGPRs (GX), dynamic LDS: 4 x 19 456 B per workgroup
        assert ring["vgpr_spill"] == 0 and ring["scratch"] == 0 and kr.waves_per_simd(ring["vgpr"]) >= 2, (inst, ring)


def test_every_structure_kernel_without_spills_is_listed_or_known():
    """Spilling instantiations are a known, short list (big structures that do not fit the register file; DESIGN section
    4): anything new on it is a regression of the generator."""
    known_big = ("l3n_mid", "l3p_", "l2p_mid", "l4n_mid", "l2n_mid", "l3n_last", "l4n_last", "l2p_second")
    spilled = {}
    current = {os.path.splitext(os.path.basename(f))[0] for f in
               glob.glob(os.path.join(ROOT, "nequip_amd", "csrc", "generated_spec", "tp_spec_*.hip"))}
    for obj in _objects("tp_spec_*.o"):
        if os.path.splitext(os.path.basename(obj))[0] not in current:
            continue  # (left over from an earlier generator run: not linked)
        for n, r in kr.kernels_of(obj).items():
            if r["vgpr_spill"] > 0:
                spilled.setdefault(os.path.basename(obj), []).append((n.split("::")[-1][:60], r["vgpr_spill"]))
    unexpected = {o: v for o, v in spilled.items() if not any(k in o for k in known_big)}
    assert not unexpected, unexpected
