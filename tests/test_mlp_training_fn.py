"""Host logic of the training-mode radial MLP Functions (nequip_amd/nn/mlp.py): with the three native launches replaced
by their float64 ATen definitions, the Function pair must reproduce first- and second-order autograd of the plain
``mm -> SiLU -> mm`` formulation (nequip/nn/mlp.py:194-196,262-268), including the parameter gradients of a
force-matching style loss.  (The launches themselves are checked on the GPU in tests/test_radial_mlp.py.)"""
import pytest
import torch

from nequip_amd.nn import mlp as M
from nequip_amd.utils import wgrad as wg


@pytest.fixture
def aten_launches(monkeypatch):
    def fwd(emb, w0, w1, a0, a1, mode, cache):
        return torch.nn.functional.silu(emb @ (w0 * a0)) @ (w1 * a1)

    def bwd(emb, w0, w1, a0, a1, g, mode, cache):
        P = emb @ (w0 * a0)
        s1, _ = M._silu_derivs(P)
        return ((g @ (w1 * a1).t()) * s1) @ (w0 * a0).t()

    def bwd_train(emb, w0, w1, a0, a1, g, cot, mode, cache):
        W0 = w0 * a0
        P = emb @ W0
        s1, s2 = M._silu_derivs(P)
        G_h = g @ (w1 * a1).t()
        if cot is None:
            G_P = G_h * s1
            return G_P @ W0.t(), torch.nn.functional.silu(P), emb.t() @ G_P
        Q = cot @ W0
        cotP = Q * G_h * s2
        return cotP @ W0.t(), Q * s1, emb.t() @ cotP + cot.t() @ (G_h * s1)

    def fwd_tangent(emb, cot, w0, w1, a0, a1, mode, cache):
        W0 = w0 * a0
        s1, _ = M._silu_derivs(emb @ W0)
        return ((cot @ W0) * s1) @ (w1 * a1)

    monkeypatch.setattr(M, "_launch_fwd", fwd)
    monkeypatch.setattr(M, "_launch_bwd", bwd)
    monkeypatch.setattr(M, "_launch_bwd_train", bwd_train)
    monkeypatch.setattr(M, "_launch_fwd_tangent", fwd_tangent)
    monkeypatch.setattr(M, "_launch_wgrad", lambda a, b: a.t() @ b)


def _loss(fn, emb, w0, w1, v, f_t, inputs_only):
    out = fn(emb, w0, w1)
    energy = (out * v).sum()
    ctx = wg.inputs_only_backward() if inputs_only else torch.enable_grad()
    with ctx:
        (force,) = torch.autograd.grad(energy, emb, create_graph=True)
    return (force - f_t).square().sum() + out.square().mean()


@pytest.mark.parametrize("mode", [M._lib.NQA_MLP_BF16X6, M._lib.NQA_MLP_FP32])
@pytest.mark.parametrize("inputs_only", [True, False])
def test_train_fn_matches_autograd_second_order(aten_launches, inputs_only, mode):
    torch.manual_seed(0)
    E, nb, H, W = 37, 8, 16, 24
    a0, a1 = 1.0 / nb**0.5, (2.0 / H) ** 0.5
    emb = torch.randn(E, nb, dtype=torch.float64, requires_grad=True)
    w0 = torch.randn(nb, H, dtype=torch.float64, requires_grad=True)
    w1 = torch.randn(H, W, dtype=torch.float64, requires_grad=True)
    v = torch.randn(E, W, dtype=torch.float64)
    f_t = torch.randn(E, nb, dtype=torch.float64)

    ref = _loss(lambda e, a, b: torch.nn.functional.silu(e @ (a * a0)) @ (b * a1), emb, w0, w1, v, f_t, False)
    g_ref = torch.autograd.grad(ref, [emb, w0, w1])
    got = _loss(lambda e, a, b: M._RadialMLPTrainFn.apply(e, a, b, a0, a1, mode, None), emb, w0, w1, v, f_t, inputs_only)
    g_got = torch.autograd.grad(got, [emb, w0, w1])
    torch.testing.assert_close(got, ref, rtol=1e-12, atol=1e-12)
    for r, g in zip(g_ref, g_got):
        torch.testing.assert_close(g, r, rtol=1e-10, atol=1e-10)


def test_silu_derivatives():
    p = torch.linspace(-6, 6, 101, dtype=torch.float64, requires_grad=True)
    y = torch.nn.functional.silu(p)
    (d1,) = torch.autograd.grad(y.sum(), p, create_graph=True)
    (d2,) = torch.autograd.grad(d1.sum(), p)
    s1, s2 = M._silu_derivs(p.detach())
    torch.testing.assert_close(s1, d1.detach(), rtol=1e-12, atol=1e-12)
    torch.testing.assert_close(s2, d2, rtol=1e-12, atol=1e-12)


def test_inputs_only_flag_is_scoped():
    assert wg.param_grads_wanted()
    with wg.inputs_only_backward():
        assert not wg.param_grads_wanted()
        with wg.inputs_only_backward():
            assert not wg.param_grads_wanted()
        assert not wg.param_grads_wanted()
    assert wg.param_grads_wanted()
