"""Polynomial cutoff envelope (mirror of ``nequip/nn/embedding/cutoffs.py:5-27``).

Only a parameter holder here: the evaluation ``1 - (p+1)(p+2)/2 x^p + p(p+2) x^(p+1) - p(p+1)/2 x^(p+2)``
masked by ``x < 1`` is fused into the edge-embedding HIP kernel (``nequip_amd/csrc/edge_embed.hip``)."""

import torch


class PolynomialCutoff(torch.nn.Module):
    def __init__(self, p: float = 6):
        super().__init__()
        assert p >= 2.0
        self.p = float(p)

    def forward(self, x: torch.Tensor) -> torch.Tensor:  # pragma: no cover
        raise RuntimeError("PolynomialCutoff is evaluated inside the fused edge-embedding HIP kernel")
