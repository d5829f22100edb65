"""Fused MFMA radial MLP (nqa_radial_mlp_fwd/bwd) against the oracle's ScalarMLPFunction restatement
(oracle/nn.py::scalar_mlp following nequip/nn/mlp.py:141-156,262-268).  float32, tolerance 1e-5 relative to
the output scale, for the GEMM modes: exact-fp32 MFMA (only the summation order differs from the CPU mm), the split-bf16
mode (three exact bf16 terms per operand, six partial products, fp32 accumulation) and the default: forward on the
two-plane fp16 split of power-of-two scaled operands (three partial products), backward on the bf16 split."""

import math

import pytest
import torch

from oracle import nn as onn


def _set_mode(monkeypatch, mode):
    monkeypatch.setenv("NQA_MLP_EXACT_FP32", "1" if mode == "fp32" else "0")
    monkeypatch.setenv("NQA_MLP_FWD_F16", "0" if mode == "bf16x6" else "1")
    monkeypatch.setenv("NQA_MLP_BWD_F16", "0" if mode == "bf16x6" else "1")


@pytest.fixture(params=["f16x3", "bf16x6", "fp32"])
def mlp_mode(request, monkeypatch):
    _set_mode(monkeypatch, request.param)
    return request.param


@pytest.mark.gpu
@pytest.mark.parametrize("E", [1, 127, 128, 1000, 4133])
@pytest.mark.parametrize("H,W", [(128, 704), (128, 192), (64, 64), (64, 160), (128, 2944), (128, 36)])
def test_radial_mlp_fwd_bwd(device, mlp_mode, E, H, W):
    from nequip_amd.nn.mlp import ScalarMLPFunction

    torch.manual_seed(E + H + W)
    mlp = ScalarMLPFunction(input_dim=8, output_dim=W, hidden_layers_depth=1, hidden_layers_width=H).eval()
    emb = torch.randn(E, 8) * 0.7
    w0, w1 = mlp.mlp[0].weight.detach(), mlp.mlp[2].weight.detach()

    e_ref = emb.clone().requires_grad_(True)
    ref = onn.scalar_mlp(e_ref, [w0, w1], "silu")
    g = torch.randn(E, W)
    (ge_ref,) = torch.autograd.grad(ref, e_ref, g)

    mlp = mlp.to(device)
    e_dev = emb.to(device).requires_grad_(True)
    assert mlp._fused_ok(e_dev), "fused MFMA path must be taken for this shape"
    out = mlp(e_dev)
    (ge,) = torch.autograd.grad(out, e_dev, g.to(device))
    torch.testing.assert_close(ref.detach(), out.detach().cpu(), atol=1e-5 * float(ref.abs().max()), rtol=1e-5)
    torch.testing.assert_close(ge_ref, ge.cpu(), atol=2e-5 * float(ge_ref.abs().max()), rtol=2e-5)


@pytest.mark.gpu
def test_radial_mlp_split_bf16_has_fp32_accuracy(device, monkeypatch):
    """Error against a float64 evaluation: the split-bf16 GEMM must be as accurate as the exact-fp32 MFMA one (and as
    a float32 CPU mm) -- i.e. at the fp32 rounding level, orders of magnitude below bf16/tf32 arithmetic."""
    from nequip_amd.nn.mlp import ScalarMLPFunction

    torch.manual_seed(7)
    E, H, W = 2048, 128, 704
    mlp = ScalarMLPFunction(input_dim=8, output_dim=W, hidden_layers_depth=1, hidden_layers_width=H).eval()
    emb = torch.randn(E, 8) * 0.7
    g = torch.randn(E, W)
    w0, w1 = mlp.mlp[0].weight.detach(), mlp.mlp[2].weight.detach()
    e64 = emb.double().requires_grad_(True)
    ref64 = onn.scalar_mlp(e64, [w0.double(), w1.double()], "silu")
    (ge64,) = torch.autograd.grad(ref64, e64, g.double())
    e32 = emb.clone().requires_grad_(True)
    ref32 = onn.scalar_mlp(e32, [w0, w1], "silu")
    (ge32,) = torch.autograd.grad(ref32, e32, g)
    err_cpu = float((ref32.detach().double() - ref64.detach()).abs().max() / ref64.abs().max())
    gerr_cpu = float((ge32.double() - ge64).abs().max() / ge64.abs().max())

    mlp = mlp.to(device)
    errs = {}
    for mode in ("fp32", "bf16x6", "f16x3"):
        _set_mode(monkeypatch, mode)
        e_dev = emb.to(device).requires_grad_(True)
        out = mlp(e_dev)
        (ge,) = torch.autograd.grad(out, e_dev, g.to(device))
        errs[mode] = (
            float((out.detach().cpu().double() - ref64.detach()).abs().max() / ref64.abs().max()),
            float((ge.cpu().double() - ge64).abs().max() / ge64.abs().max()),
        )
    print("max error / max|ref| vs float64: cpu fp32", (err_cpu, gerr_cpu), errs)
    for mode, (ef, eb) in errs.items():
        assert ef < 2e-6 and eb < 4e-6, (mode, ef, eb)
    # the split modes within 3x of the exact-fp32 kernels' own rounding error
    for mode in ("bf16x6", "f16x3"):
        assert errs[mode][0] < 3 * max(errs["fp32"][0], err_cpu), mode
        assert errs[mode][1] < 3 * max(errs["fp32"][1], gerr_cpu), mode


@pytest.mark.gpu
def test_radial_mlp_f16x3_forward_is_scale_robust(device, monkeypatch):
    """fp16 has five exponent bits: the forward scales every hidden row and every 32-column weight tile by a power of
    two before splitting it.  Rows and column tiles whose magnitudes differ by many orders (embedding rows down to 1e-12
    of the largest -- edges at the cutoff --, a zero row, weight columns from 1e-6 to 1e+5) must come out at the fp32
    level RELATIVE TO THEIR OWN row x tile scale, like the bf16 split (fp32 exponent range) and the exact mode."""
    from nequip_amd.nn.mlp import ScalarMLPFunction

    torch.manual_seed(11)
    E, H, W = 640, 128, 256
    mlp = ScalarMLPFunction(input_dim=8, output_dim=W, hidden_layers_depth=1, hidden_layers_width=H).eval()
    with torch.no_grad():
        col = torch.ones(W)
        col[32:64] = 1e-6
        col[64:96] = 1e5
        col[96:128] = 3e-3
        col[130] = 1e-4  # one small column inside an O(1) tile: precision relative to the tile, not to itself
        mlp.mlp[2].weight.mul_(col)
    emb = torch.randn(E, 8) * 0.7
    row = torch.ones(E)
    row[100:200] = 1e-4
    row[200:300] = 1e-8
    row[300:400] = 1e-12
    row[400:420] = 30.0
    emb = emb * row[:, None]
    emb[7] = 0.0
    w0, w1 = mlp.mlp[0].weight.detach(), mlp.mlp[2].weight.detach()
    ref = onn.scalar_mlp(emb.double(), [w0.double(), w1.double()], "silu")
    # scale of (row, tile): largest |ref| of the row within the 32-column tile
    scale = ref.abs().reshape(E, W // 32, 32).amax(dim=2, keepdim=True).expand(E, W // 32, 32).reshape(E, W)
    scale = scale.clamp_min(1e-300)
    mlp = mlp.to(device)
    errs = {}
    for mode in ("fp32", "bf16x6", "f16x3"):
        _set_mode(monkeypatch, mode)
        out = mlp(emb.to(device)).cpu().double()
        assert torch.isfinite(out).all(), mode
        assert torch.equal(out[7], torch.zeros(W, dtype=torch.float64)), mode
        errs[mode] = float(((out - ref).abs() / scale).max())
    print("max error relative to the (row, tile) scale:", errs)
    assert errs["f16x3"] < 3e-6 and errs["bf16x6"] < 3e-6 and errs["fp32"] < 3e-6, errs
    assert errs["f16x3"] < 3 * max(errs["fp32"], errs["bf16x6"]) + 5e-7, errs


@pytest.mark.gpu
@pytest.mark.parametrize("paired", [False, True])
def test_radial_mlp_f16x3_backward_running_scale(device, monkeypatch, paired):
    """The backward streams the gradient rows chunk by chunk with no scale known in advance: the fp16 split carries a
    running per-row exponent and rescales a row's accumulators when a later chunk outgrows it.  Rows whose chunks
    differ by 20 orders of magnitude in either direction (small first / large first), rows of 1e-25 and 1e+12, a zero
    row, and weight column groups from 1e-6 to 1e+5: the error of every row stays at the fp32 level relative to the
    magnitude sum  |g| |W1| |silu'| |W0|  of that row (what an fp32 evaluation is accurate to), like the bf16 split."""
    from nequip_amd.nn import mlp as m

    torch.manual_seed(5)
    E, H, W = 384, 128, 320
    mod = m.ScalarMLPFunction(input_dim=8, output_dim=W, hidden_layers_depth=1, hidden_layers_width=H).eval()
    with torch.no_grad():
        col = torch.ones(W)
        col[32:64] = 1e-6
        col[96:128] = 1e5
        col[200] = 1e-3
        mod.mlp[2].weight.mul_(col)
    emb = torch.randn(E, 8) * 0.7
    g = torch.randn(E, W)
    g[10:40, :64] *= 1e-20          # small chunks first, then O(1): rescale on the third chunk
    g[40:70, 64:] *= 1e-20          # large chunks first: the tail must not be lost relative to the head's scale
    g[70:100] *= 1e-25
    g[100:130] *= 1e12
    for r in range(130, 160):       # staircase: every chunk 30x the previous one
        for c in range(W // 32):
            g[r, 32 * c:32 * c + 32] *= 30.0 ** c * 1e-8
    g[7] = 0.0
    w0, w1 = mod.mlp[0].weight.detach(), mod.mlp[2].weight.detach()
    a0, a1 = 1.0 / math.sqrt(8), math.sqrt(2.0) / math.sqrt(H)
    e64 = emb.double().requires_grad_(True)
    ref = onn.scalar_mlp(e64, [w0.double(), w1.double()], "silu")
    (ge64,) = torch.autograd.grad(ref, e64, g.double())
    pre = emb.double() @ (w0.double() * a0)
    sig = torch.sigmoid(pre)
    dsilu = sig * (1 + pre * (1 - sig))
    denom = ((g.double().abs() @ (w1.double() * a1).abs().T) * dsilu.abs()) @ (w0.double() * a0).abs().T
    denom = denom.amax(dim=1, keepdim=True).clamp_min(1e-300)
    mod = mod.to(device)
    cache = m._WeightImages()
    cache.validate(mod.mlp[2].weight)
    errs = {}
    for mode in ("fp32", "bf16x6", "f16x3"):
        _set_mode(monkeypatch, mode)
        args = (emb.to(device), mod.mlp[0].weight.detach(), mod.mlp[2].weight.detach(), a0, a1)
        if paired:  # the same gradient as two row streams that the kernel adds while loading
            if mode == "fp32":
                continue
            g_a = (g * 0.25).to(device)
            ge = m._launch_bwd_paired(*args, g_a, (g - g * 0.25).to(device), m.radial_mlp_mode(), cache)
        else:
            ge = m._launch_bwd(*args, g.to(device), m.radial_mlp_mode(), cache)
        ge = ge.cpu().double()
        assert torch.isfinite(ge).all(), mode
        assert torch.equal(ge[7], torch.zeros(8, dtype=torch.float64)), mode
        errs[mode] = float(((ge - ge64).abs() / denom).max())
    print("max row error / row magnitude sum:", errs)
    for mode, e in errs.items():
        assert e < 1e-6, errs
    assert errs["f16x3"] < 3 * errs["bf16x6"] + 2e-7, errs


def _force_matching_loss(mlp_fn, emb, v, f_t):
    out = mlp_fn(emb)
    (force,) = torch.autograd.grad((out * v).sum(), emb, create_graph=True)
    return (force - f_t).square().sum() + out.square().sum() / max(1, out.numel())


@pytest.mark.gpu
@pytest.mark.parametrize("E,H,W", [(1000, 128, 192), (4133, 128, 704), (77, 64, 64), (0, 128, 192)])
def test_radial_mlp_training_mode_second_order(device, mlp_mode, E, H, W):
    """Training mode: fused kernels inside the twice-differentiable Function pair.  Parameter and input gradients of a
    force-matching style loss (first derivative inside the loss) against float64 autograd of the oracle's restatement."""
    from nequip_amd.nn.mlp import ScalarMLPFunction, _RadialMLPTrainFn

    torch.manual_seed(E + W)
    mlp = ScalarMLPFunction(input_dim=8, output_dim=W, hidden_layers_depth=1, hidden_layers_width=H).train()
    emb = torch.randn(E, 8) * 0.7
    v, f_t = torch.randn(E, W), torch.randn(E, 8)

    w_ref = [mlp.mlp[0].weight.detach().double().requires_grad_(True),
             mlp.mlp[2].weight.detach().double().requires_grad_(True)]
    e_ref = emb.double().requires_grad_(True)
    loss_ref = _force_matching_loss(lambda e: onn.scalar_mlp(e, w_ref, "silu"), e_ref, v.double(), f_t.double())
    g_ref = torch.autograd.grad(loss_ref, [e_ref] + w_ref)

    mlp = mlp.to(device)
    e_dev = emb.to(device).requires_grad_(True)
    assert mlp._fused_ok(e_dev)
    seen = []
    orig = _RadialMLPTrainFn.forward
    try:
        _RadialMLPTrainFn.forward = staticmethod(lambda *a, **k: (seen.append(1), orig(*a, **k))[1])
        loss = _force_matching_loss(mlp, e_dev, v.to(device), f_t.to(device))
    finally:
        _RadialMLPTrainFn.forward = staticmethod(orig)
    assert seen, "training mode must run the fused Function pair for this shape"
    loss.backward()
    got = [e_dev.grad, mlp.mlp[0].weight.grad, mlp.mlp[2].weight.grad]
    torch.testing.assert_close(loss.detach().cpu().double(), loss_ref.detach(), atol=1e-4 * max(1.0, float(loss_ref)), rtol=1e-4)
    for r, g in zip(g_ref, got):
        assert g is not None
        torch.testing.assert_close(g.cpu().double(), r, atol=2e-4 * max(1e-6, float(r.abs().max())) if r.numel() else 0.0,
                                   rtol=1e-3)


@pytest.mark.gpu
def test_radial_mlp_training_aten_switch(device, monkeypatch):
    """NQA_MLP_TRAIN_ATEN=1 keeps the mm/SiLU formulation in training mode (debugging switch); both give parameter
    gradients."""
    from nequip_amd.nn.mlp import ScalarMLPFunction

    mlp = ScalarMLPFunction(input_dim=8, output_dim=192, hidden_layers_depth=1, hidden_layers_width=128).to(device)
    mlp.train()
    x = torch.randn(300, 8, device=device)
    mlp(x).square().sum().backward()
    fused = [p.grad.clone() for p in mlp.parameters()]
    monkeypatch.setenv("NQA_MLP_TRAIN_ATEN", "1")
    mlp.zero_grad()
    mlp(x).square().sum().backward()
    for a, b in zip(fused, [p.grad for p in mlp.parameters()]):
        assert torch.isfinite(a).all()
        torch.testing.assert_close(a, b, atol=1e-4 * float(b.abs().max()), rtol=1e-3)


@pytest.mark.gpu
@pytest.mark.parametrize("depth", [1, 2, 3])
def test_radial_mlp_zero_edges_first_then_real_graph(device, depth):
    """A first call on an empty edge list (isolated atom) must not leave a never-filled weight image behind that the next
    call with edges would take for a filled one (``_WeightImages.get(..., rows)``): every launch of the fused path, the
    ``nqa_radial_mlp_last_*`` layers of a deeper MLP included, passes its row count."""
    from nequip_amd.nn.mlp import ScalarMLPFunction

    torch.manual_seed(11 + depth)
    H, W, E = 64, 160, 777
    mlp = ScalarMLPFunction(input_dim=8, output_dim=W, hidden_layers_depth=depth, hidden_layers_width=H).eval()
    ws = [m.weight.detach().clone() for m in mlp.mlp if hasattr(m, "weight")]
    emb = torch.randn(E, 8) * 0.7
    e_ref = emb.clone().requires_grad_(True)
    ref = onn.scalar_mlp(e_ref, ws, "silu")
    g = torch.randn(E, W)
    (ge_ref,) = torch.autograd.grad(ref, e_ref, g)

    mlp = mlp.to(device)
    e0 = torch.zeros(0, 8, device=device, requires_grad=True)
    out0 = mlp(e0)
    assert out0.shape == (0, W)
    (g0,) = torch.autograd.grad(out0, e0, torch.zeros(0, W, device=device))
    assert g0.shape == (0, 8)
    e_dev = emb.to(device).requires_grad_(True)
    out = mlp(e_dev)
    (ge,) = torch.autograd.grad(out, e_dev, g.to(device))
    torch.testing.assert_close(ref.detach(), out.detach().cpu(), atol=1e-5 * float(ref.abs().max()), rtol=1e-5)
    torch.testing.assert_close(ge_ref, ge.cpu(), atol=2e-5 * float(ge_ref.abs().max()), rtol=2e-5)


# Kernel variants that are not the default (measured slower in the step) but stay selectable: every one
# of them has to give the default's results.  The switches are read at every call.
MLP_VARIANTS = {
    "general_fwd_bwd": {"NQA_MLP_PIPE": "0"},
    "fwd_direct_epilogue": {"NQA_MLP_PIPE_DIRECT": "1"},
    "fwd_one_workgroup_per_block": {"NQA_MLP_FWD_BALANCED": "0"},
    "bwd_coalesced_rows": {"NQA_MLP_BWD_COAL": "1"},
    "bwd_balanced_ranges": {"NQA_MLP_BWD_BALANCED": "1"},
    "bwd_balanced_prefetch4": {"NQA_MLP_BWD_BALANCED": "1", "NQA_MLP_BWD_PF": "4"},
    "bwd_small_lds_resident": {"NQA_MLP_BWD_SMALL": "2"},
    "bwd_half_chip": {"NQA_MLP_BWD_COAL": "1", "NQA_MLP_BWD_WGS_PER_CU": "1"},
}


@pytest.mark.gpu
@pytest.mark.parametrize("variant", list(MLP_VARIANTS))
@pytest.mark.parametrize("E,W", [(4133, 704), (70001, 192), (513, 64)])
def test_radial_mlp_opt_in_kernels_match_the_default(device, monkeypatch, variant, E, W):
    from nequip_amd.nn.mlp import ScalarMLPFunction

    _set_mode(monkeypatch, "f16x3")
    torch.manual_seed(E + W)
    mlp = ScalarMLPFunction(input_dim=8, output_dim=W, hidden_layers_depth=1, hidden_layers_width=128).eval().to(device)
    emb = (torch.randn(E, 8) * 0.7).to(device)
    g = torch.randn(E, W).to(device)

    def run():
        e = emb.clone().requires_grad_(True)
        out = mlp(e)
        (ge,) = torch.autograd.grad(out, e, g)
        return out.detach(), ge

    ref_out, ref_ge = run()
    for k_, v_ in MLP_VARIANTS[variant].items():
        monkeypatch.setenv(k_, v_)
    out, ge = run()
    # (different summation orders of the same split products: equal at the fp32 rounding level)
    torch.testing.assert_close(out, ref_out, atol=2e-6 * float(ref_out.abs().max()), rtol=2e-6)
    torch.testing.assert_close(ge, ref_ge, atol=5e-6 * float(ref_ge.abs().max()), rtol=5e-6)
