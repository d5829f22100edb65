"""HIP path against the committed golden fixtures (tests/golden/*.npz, produced by tests/golden/make_golden.py
from the oracle): float64, tolerance 1e-10 -- independent of /root/reference and of the oracle at run time."""

import os

import numpy as np
import pytest
import torch

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.gpu
def test_tp_scatter_golden(device):
    from nequip_amd.model.nequip_models import torch_default_dtype
    from nequip_amd.nn import TensorProductScatter
    from nequip_amd.o3 import Irreps

    gold = np.load(os.path.join(GOLDEN, "tp_scatter.npz"))
    t = lambda k: torch.from_numpy(gold[k]).to(device)  # noqa: E731
    instr = [(int(a), int(b), int(c), "uvu", True) for a, b, c in gold["instructions"]]
    with torch_default_dtype(torch.float64):
        tps = TensorProductScatter(Irreps(str(gold["feature_irreps_in"])), Irreps(str(gold["irreps_edge_attr"])),
                                   Irreps(str(gold["irreps_mid"])), instr).to(device)
    x, y, w = t("x").requires_grad_(True), t("y").requires_grad_(True), t("w").requires_grad_(True)
    out = tps(x, y, w, t("dst"), t("src"))
    np.testing.assert_allclose(out.detach().cpu().numpy(), gold["out"], atol=1e-10)
    gx, gy, gw = torch.autograd.grad(out, [x, y, w], t("go"))
    np.testing.assert_allclose(gx.cpu().numpy(), gold["gx"], atol=1e-10)
    np.testing.assert_allclose(gy.cpu().numpy(), gold["gy"], atol=1e-10)
    np.testing.assert_allclose(gw.cpu().numpy(), gold["gw"], atol=1e-10)


@pytest.mark.gpu
def test_edge_embed_golden(device):
    from nequip_amd.nn.embedding._edge import _EdgeEmbedFn

    gold = np.load(os.path.join(GOLDEN, "edge_embed.npz"))
    vec = torch.from_numpy(gold["vec"]).to(device)
    cfg = dict(dtype=torch.float64, lmax=4, want_sh=True, want_emb=True, nb=8, rmax_recip=1.0 / 4.5, p=6.0,
               factor=2 * np.pi / 4.5**2)
    sh, emb = _EdgeEmbedFn.apply(vec, torch.linspace(1.0, 8.0, 8, dtype=torch.float64, device=device), cfg)
    np.testing.assert_allclose(sh.cpu().numpy(), gold["sh"], atol=1e-12)
    np.testing.assert_allclose(emb.cpu().numpy(), gold["emb"], atol=1e-12)


@pytest.mark.gpu
def test_model_golden_float64(device):
    from nequip_amd.model import NequIPGNNModel

    gold = np.load(os.path.join(GOLDEN, "model_si64.npz"))
    model = NequIPGNNModel(seed=3, model_dtype="float64", type_names=["Si"], r_max=4.5, num_layers=3, l_max=2,
                           parity=False, num_features=8, radial_mlp_depth=1, radial_mlp_width=16, num_bessels=8,
                           polynomial_cutoff_p=6, avg_num_neighbors=20.0)
    sd = {"model.func." + k[3:]: torch.from_numpy(gold[k]) for k in gold.files if k.startswith("w::")}
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected
    model = model.to(device).eval()
    data = {"pos": torch.from_numpy(gold["pos"]), "atom_types": torch.from_numpy(gold["types"]),
            "edge_index": torch.from_numpy(gold["edge_index"]), "cell": torch.from_numpy(gold["cell"]).view(1, 3, 3),
            "edge_cell_shift": torch.from_numpy(gold["edge_cell_shift"])}
    out = model({k: v.to(device) for k, v in data.items()})
    np.testing.assert_allclose(out["total_energy"].detach().cpu().numpy(), gold["total_energy"], atol=1e-9)
    np.testing.assert_allclose(out["forces"].cpu().numpy(), gold["forces"], atol=1e-9)
    np.testing.assert_allclose(out["virial"].cpu().numpy(), gold["virial"], atol=1e-8)
