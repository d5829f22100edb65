"""The arithmetic scheme behind the fp16 modes of the radial MLP and the node linears, restated on the CPU (torch.half
conversions are IEEE round-to-nearest-even like v_cvt_f16_f32) -- the kernels' accuracy claims, checked without a GPU:

* two planes  h = fp16(x), l = fp16(x - h)  carry x to 2^-22 |x|; three products h h + h l + l h reproduce a dot product
  to that level (``radial_mlp.hip``: split_pair_f16);
* operands are first multiplied by the power of two that puts their group's largest magnitude into [2^14, 2^15)
  (f16_scale_up); whatever then falls below fp16's smallest normal number costs at most 2^-28 of the group's maximum,
  with subnormal inputs kept OR flushed to zero by the matrix pipe (both emulated);
* a streamed operand without a known scale carries a running exponent that is lowered -- with the accumulator -- when a
  chunk outgrows it (the backward of the radial MLP, the node linears): exact powers of two, so the result only carries
  the split's own error, relative to (running row maximum) x (chunk weight maximum).
"""
import math

import pytest
import torch

F16_MIN_NORMAL = 2.0 ** -14


def scale_up_exponent(m: float) -> int:
    """k with m 2^k in [2^14, 2^15) (0 for m = 0 / non-finite; clamped like the kernels)."""
    if not (m > 0.0) or not math.isfinite(m):
        return 0
    _, e = math.frexp(m)  # m = f 2^e, f in [0.5, 1)
    return max(-100, min(100, 15 - e))


def split(x: torch.Tensor, flush: bool):
    """(h, l) as float64 values of the two fp16 planes; flush=True zeroes subnormal plane values (a pipe that flushes)."""
    x = x.float()
    h = x.half()
    l = (x - h.float()).half()
    h, l = h.double(), l.double()
    if flush:
        h = torch.where(h.abs() < F16_MIN_NORMAL, torch.zeros_like(h), h)
        l = torch.where(l.abs() < F16_MIN_NORMAL, torch.zeros_like(l), l)
    return h, l


def dot3(xh, xl, wh, wl):
    return xh @ wh + xh @ wl + xl @ wh


def test_two_planes_carry_22_bits():
    g = torch.Generator().manual_seed(0)
    mant = 1.0 + torch.rand(100000, generator=g)
    sign = torch.where(torch.rand(100000, generator=g) < 0.5, -1.0, 1.0)
    # 2^-3 <= |x| < 2^15 with subnormal plane values kept: 22 bits relative
    x = sign * mant * 2.0 ** torch.randint(-3, 15, (100000,), generator=g).float()
    h, l = split(x, flush=False)
    assert float((((h + l) - x.double()).abs() / x.double().abs()).max()) <= 2.0 ** -22
    # any magnitude below 2^15: the error is at most 2^-11 of the low part where that is a normal number, and otherwise
    # absolute -- fp16's smallest normal number 2^-14 if the pipe flushes subnormal inputs, half a subnormal step 2^-25 if
    # it keeps them: 2^-28 / 2^-39 of a group maximum of 2^14
    x = torch.cat([x, sign * mant * 2.0 ** torch.randint(-30, -3, (100000,), generator=g).float()])
    for flush, floor in ((True, 2.0 ** -14), (False, 2.0 ** -25)):
        h, l = split(x, flush)
        err = ((h + l) - x.double()).abs()
        assert bool((err <= torch.maximum(x.double().abs() * 2.0 ** -22, torch.tensor(floor, dtype=torch.float64))).all())


@pytest.mark.parametrize("flush", [False, True])
def test_scaled_split_dot_products_are_fp32_accurate(flush):
    """Forward scheme: rows of x and column tiles of w scaled by their own power of two.  Rows from 1e-12 to 1e+8, weight
    tiles from 1e-6 to 1e+5, elements far below their group's maximum inside every group."""
    g = torch.Generator().manual_seed(1)
    P, K, W = 64, 128, 96
    x = torch.randn(P, K, generator=g)
    x *= (10.0 ** torch.linspace(-12, 8, P))[:, None]
    x[:, ::7] *= 1e-9        # elements 2^-30 of their row's maximum: lost, at no visible cost
    w = torch.randn(K, W, generator=g)
    w[:, 32:64] *= 1e-6
    w[:, 64:] *= 1e5
    w[::5] *= 1e-7
    x, w = x.float(), w.float()
    exact = x.double() @ w.double()
    mag = x.double().abs() @ w.double().abs()
    out = torch.zeros_like(exact)
    for p in range(P):
        kx = scale_up_exponent(float(x[p].abs().max()))
        xh, xl = split(torch.ldexp(x[p], torch.tensor(kx)), flush)
        for t in range(W // 32):
            wt = w[:, 32 * t:32 * t + 32]
            kw = scale_up_exponent(float(wt.abs().max()))
            wh, wl = split(torch.ldexp(wt, torch.tensor(kw)), flush)
            out[p, 32 * t:32 * t + 32] = dot3(xh[None], xl[None], wh, wl)[0] * 2.0 ** (-kx - kw)
    # every element at the split's level relative to (row maximum) x (tile maximum) x K, i.e. far below the fp32 rounding
    # of its magnitude sum except where the sum itself is dominated by the lost small elements
    bound = (x.double().abs().amax(dim=1, keepdim=True) * 2.0 ** -21) * torch.stack(
        [w[:, 32 * t:32 * t + 32].double().abs().amax() for t in range(W // 32)]).repeat_interleave(32)[None, :] * K
    assert bool(((out - exact).abs() <= bound).all())
    typical = ((out - exact).abs() / mag).median()
    assert float(typical) < 2.0 ** -22


@pytest.mark.parametrize("flush", [False, True])
def test_running_row_exponent_tracks_a_streamed_operand(flush):
    """Backward scheme: g arrives in chunks of 32 columns; each row keeps S, chunk c is multiplied by 2^(S - e_c) with e_c
    the exponent its weight chunk was scaled by, S (and the accumulator) drop when a chunk would leave fp16's range."""
    g_ = torch.Generator().manual_seed(2)
    P, W, H = 48, 320, 64
    grad = torch.randn(P, W, generator=g_)
    grad[0:8, :64] *= 1e-20                     # small chunks first
    grad[8:16, 64:] *= 1e-20                    # large chunks first
    for c in range(W // 32):                    # staircase: a rescale at every chunk
        grad[16:24, 32 * c:32 * c + 32] *= 30.0 ** c * 1e-8
    grad[24:32] *= 1e-25
    grad[32:40] *= 1e12
    grad[40] = 0.0
    w = torch.randn(W, H, generator=g_)        # B[k][j]
    w[32:64] *= 1e-6
    w[96:128] *= 1e5
    grad, w = grad.float(), w.float()
    exact = grad.double() @ w.double()
    mag = grad.double().abs() @ w.double().abs()
    out = torch.zeros_like(exact)
    rescales = 0
    for p in range(P):
        S, acc = None, torch.zeros(H, dtype=torch.float64)
        for c in range(W // 32):
            wc = w[32 * c:32 * c + 32]
            e_c = scale_up_exponent(float(wc.abs().max()))
            wh, wl = split(torch.ldexp(wc, torch.tensor(e_c)), flush)
            gc = grad[p, 32 * c:32 * c + 32]
            m = float(gc.abs().max())
            if m > 0 and math.isfinite(m):
                _, em = math.frexp(m)
                cap = 15 - em + e_c
                if S is None or cap < S:
                    ns = max(e_c - 100, min(e_c + 100, cap - 3))
                    if S is not None:
                        acc = acc * 2.0 ** (ns - S)
                        rescales += 1
                    S = ns
            q = 0 if S is None else max(-120, min(120, S - e_c))
            gh, gl = split(torch.ldexp(gc, torch.tensor(q)), flush)
            assert float(gh.abs().max()) < 2.0 ** 15
            acc = acc + dot3(gh[None], gl[None], wh, wl)[0]
        out[p] = acc * 2.0 ** (-(S or 0))
    assert rescales >= 8 * (W // 32 - 2)  # the staircase rows rescale at (almost) every chunk
    assert torch.equal(out[40], torch.zeros(H, dtype=torch.float64))
    err = (out - exact).abs() / mag.clamp_min(1e-300)
    assert float(err.max()) < 2e-6 and float(err.median()) < 2.0 ** -21
