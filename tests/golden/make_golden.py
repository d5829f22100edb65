#!/usr/bin/env python3
"""Generate the committed golden fixtures from the CPU oracle (``oracle/``).

The reference cannot be imported here (``e3nn`` is absent) and ships no golden vectors for this
path, so these fixtures are *oracle outputs*: they pin the oracle (and through it the HIP kernels) against
regressions and travel to the GPU box, but they do not pin parity with e3nn ("parity unpinned").

    python tests/golden/make_golden.py        # rewrites tests/golden/*.npz (float64 arrays, seeded inputs)
"""

import os
import sys

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)

from oracle import model as omodel  # noqa: E402
from oracle import nn as onn  # noqa: E402
from oracle import tp as otp  # noqa: E402
from oracle.sh import spherical_harmonics  # noqa: E402
from oracle.wigner import wigner_3j  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def golden_wigner():
    out = {}
    for l1 in range(4):
        for l2 in range(4):
            for l3 in range(abs(l1 - l2), min(l1 + l2, 3) + 1):
                out[f"w3j_{l1}_{l2}_{l3}"] = wigner_3j(l1, l2, l3).numpy()
    np.savez_compressed(os.path.join(HERE, "wigner_3j.npz"), **out)


def golden_edge_embed():
    g = torch.Generator().manual_seed(20260923)
    vec = torch.randn(64, 3, generator=g, dtype=torch.float64) * 2.0
    vec[0] = torch.tensor([0.0, 5.0, 0.0])  # beyond r_max = 4.5
    vec[1] = torch.tensor([0.0, 0.0, 1.0])
    sh = spherical_harmonics(vec, 4)
    emb, cut = onn.bessel_embedding(vec, 4.5, 8, 6.0, torch.float64)
    np.savez_compressed(os.path.join(HERE, "edge_embed.npz"), vec=vec.numpy(), sh=sh.numpy(), emb=emb.numpy(),
                        cutoff=cut.numpy())


def golden_tp_scatter():
    g = torch.Generator().manual_seed(7)
    f_in, e_at, filt = "8x0e + 8x2e + 8x1o", "0e + 1o + 2e", "24x0e + 32x1o + 16x1e + 16x2o + 32x2e"
    mid, instr = otp.build_instructions(f_in, e_at, filt)
    from oracle import irreps as ir

    N, E = 8, 15
    W = otp.weight_numel(f_in, e_at, instr)
    x = torch.randn(N, ir.dim(ir.parse(f_in)), generator=g, dtype=torch.float64)
    y = torch.randn(E, 9, generator=g, dtype=torch.float64)
    w = torch.randn(E, W, generator=g, dtype=torch.float64)
    src = torch.randint(0, N, (E,), generator=g)
    dst = torch.randint(0, N, (E,), generator=g)
    x.requires_grad_(True), y.requires_grad_(True), w.requires_grad_(True)
    out = otp.tp_scatter(x, y, w, dst, src, f_in, e_at, ir.to_str(mid), instr)
    go = torch.randn(out.shape, generator=g, dtype=torch.float64)
    gx, gy, gw = torch.autograd.grad(out, [x, y, w], go)
    np.savez_compressed(
        os.path.join(HERE, "tp_scatter.npz"), x=x.detach().numpy(), y=y.detach().numpy(), w=w.detach().numpy(),
        src=src.numpy(), dst=dst.numpy(), out=out.detach().numpy(), go=go.numpy(), gx=gx.numpy(), gy=gy.numpy(),
        gw=gw.numpy(), instructions=np.array([[a, b, c] for a, b, c, *_ in instr]), irreps_mid=ir.to_str(mid),
        feature_irreps_in=f_in, irreps_edge_attr=e_at,
    )


def golden_model():
    """Tiny NequIP model (l_max 2, parity False, 8 features, 3 layers) on a rattled 2x2x2 Si cell: float64."""
    from nequip_amd.model import NequIPGNNModel
    from nequip_amd.utils import synthetic as syn

    pos, types, cell, names = syn.silicon_box(reps=2, seed=11)
    data = syn.make_data(pos, types, 4.5, cell)
    cfg = dict(r_max=4.5, num_layers=3, l_max=2, parity=False, num_features=8, radial_mlp_depth=1,
               radial_mlp_width=16, num_bessels=8, polynomial_cutoff_p=6, avg_num_neighbors=20.0,
               model_dtype="float64")
    model = NequIPGNNModel(seed=3, model_dtype="float64", type_names=names,
                           **{k: v for k, v in cfg.items() if k != "model_dtype"})
    weights = {k.replace("model.func.", ""): v.detach() for k, v in model.state_dict().items()}
    ref = omodel.energy_forces(data, cfg, weights, with_virial=True)
    np.savez_compressed(
        os.path.join(HERE, "model_si64.npz"), pos=pos, types=types, cell=cell,
        edge_index=data["edge_index"].numpy(), edge_cell_shift=data["edge_cell_shift"].numpy(),
        total_energy=ref["total_energy"].numpy(), forces=ref["forces"].numpy(), virial=ref["virial"].numpy(),
        **{"w::" + k: v.numpy() for k, v in weights.items()},
    )


if __name__ == "__main__":
    golden_wigner()
    golden_edge_embed()
    golden_tp_scatter()
    golden_model()
    print(sorted(os.listdir(HERE)))
