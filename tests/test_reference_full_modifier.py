"""`enable_NequipAMD_full` on a model that the REFERENCE's own builder produced, applied by the REFERENCE's own `modify`
(runs where /root/reference exists, i.e. in the build container; the GPU box has no reference and skips).

nequip is imported from /root/reference.  e3nn is not installable here, so the e3nn classes nequip's modules construct are
served by stand-ins -- but functional ones this time: `e3nn.o3.Irreps / Irrep`, `Linear`, `FullyConnectedTensorProduct`,
`TensorProduct`, `Gate`, `NormActivation` resolve to this package's CPU mirrors (the Gate re-shaped like e3nn's: its
activations sit in `Activation`-like holders of `normalize2mom`-like wrappers), everything else of e3nn / the training stack
to the inert objects of tests/golden/make_reference_golden.py.  What runs is nequip's real code: `nequip.model.NequIPGNNModel`
(`nequip/model/nequip_models.py:116-399` under the `model_builder` wrapper, `nequip/model/utils.py:104-216`), every
`nequip.nn` constructor on the way, modifier discovery and `nequip.model.modify` (`nequip/model/modify_utils.py:35-150`).
Checked: the converted tree is module for module the one `nequip_amd.model.NequIPGNNModel` builds for the same
hyper-parameters (the model bench.py times), with the reference model's parameters, the same cross-module fusion plan, and an
unchanged set of state-dict keys.
"""

import contextlib
import importlib.util
import os
import sys
import warnings

import pytest
import torch

REFERENCE = os.environ.get("NEQUIP_REFERENCE", "/root/reference")
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REFERENCE, "nequip")), reason="reference tree not present")

HYPER = dict(seed=3, model_dtype="float32", type_names=["H", "O"], r_max=4.5, num_layers=3, l_max=2, parity=False,
             num_features=8, radial_mlp_width=16, avg_num_neighbors=20.0,
             per_type_energy_scales={"H": 1.5, "O": 0.75}, per_type_energy_shifts={"H": -1.0, "O": -3.0})


@pytest.fixture(scope="module")
def ref():
    spec = importlib.util.spec_from_file_location(
        "make_reference_golden", os.path.join(os.path.dirname(__file__), "golden", "make_reference_golden.py"))
    mrg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mrg)

    from nequip_amd.o3 import irreps as my_irreps, modules as my_modules, tensor_product as my_tp

    class _Normalize2Mom(torch.nn.Module):  # e3nn.math.normalize2mom: the wrapped function is `.f`
        def __init__(self, f):
            super().__init__()
            self.f = f

    class _Activation(torch.nn.Module):  # e3nn.nn.Activation: `.acts` holds the wrappers
        def __init__(self, acts):
            super().__init__()
            self.acts = torch.nn.ModuleList([_Normalize2Mom(a) for a in acts])

    class Gate(my_modules.Gate):
        def __init__(self, irreps_scalars, act_scalars, irreps_gates, act_gates, irreps_gated):
            super().__init__(irreps_scalars, act_scalars, irreps_gates, act_gates, irreps_gated)
            self.act_scalars = _Activation(act_scalars)
            self.act_gates = _Activation(act_gates)

    class SphericalHarmonics(torch.nn.Module):
        def __init__(self, irreps_out, normalize, normalization="integral", irreps_in=None):
            super().__init__()
            self.irreps_out, self.normalize, self.normalization = my_irreps.Irreps(str(irreps_out)), normalize, normalization

    class Linear(my_modules.Linear):
        def __init__(self, irreps_in, irreps_out, internal_weights=True, shared_weights=True, **kw):
            super().__init__(irreps_in=irreps_in, irreps_out=irreps_out)

    class TensorProduct(my_tp.TensorProduct):
        def __init__(self, irreps_in1, irreps_in2, irreps_out, instructions, shared_weights=False, internal_weights=False, **kw):
            super().__init__(irreps_in1, irreps_in2, irreps_out, instructions)

    @contextlib.contextmanager
    def isolate_rng(*a, **k):
        state = torch.get_rng_state()
        try:
            yield
        finally:
            torch.set_rng_state(state)

    special = {
        "e3nn.o3._irreps": dict(Irreps=my_irreps.Irreps, Irrep=my_irreps.Irrep),
        "e3nn.o3": dict(Irreps=my_irreps.Irreps, Irrep=my_irreps.Irrep),
        "e3nn.o3._linear": dict(Linear=Linear),
        "e3nn.o3._tensor_product._sub": dict(FullyConnectedTensorProduct=my_modules.FullyConnectedTensorProduct),
        "e3nn.o3._tensor_product._tensor_product": dict(TensorProduct=TensorProduct),
        "e3nn.nn._gate": dict(Gate=Gate),
        "e3nn.nn._normact": dict(NormActivation=my_modules.NormActivation),
        "e3nn.o3._spherical_harmonics": dict(SphericalHarmonics=SphericalHarmonics),
        "e3nn.util.jit": dict(compile_mode=lambda mode: (lambda cls: cls)),
        "lightning.pytorch.utilities.seed": dict(isolate_rng=isolate_rng),
    }

    class Finder(mrg._Finder):
        def exec_module(self, module):
            for k, v in special.get(module.__name__, {}).items():
                setattr(module, k, v)

    finder = Finder()
    for k in [k for k in sys.modules if k == "nequip" or k.startswith("nequip.")]:
        del sys.modules[k]
    sys.meta_path.insert(0, finder)
    sys.path.insert(0, REFERENCE)
    real_filterwarnings = warnings.filterwarnings
    warnings.filterwarnings = lambda action, message="", category=Warning, *a, **k: (
        real_filterwarnings(action, message, category, *a, **k) if isinstance(category, type) else None)
    # nequip's set_global_state switches process-wide torch settings (default dtype float64, TF32 flags): put them back
    saved = (torch.get_default_dtype(), torch.backends.cuda.matmul.allow_tf32, torch.backends.cudnn.allow_tf32)
    try:
        import nequip.model as nm
        import nequip.model.modify_utils as modify_utils
        import nequip.nn as rnn
        from nequip.utils.global_state import set_global_state

        set_global_state()
        yield dict(model=nm, modify_utils=modify_utils, nn=rnn)
    finally:
        torch.set_default_dtype(saved[0])
        torch.backends.cuda.matmul.allow_tf32, torch.backends.cudnn.allow_tf32 = saved[1], saved[2]
        warnings.filterwarnings = real_filterwarnings
        sys.meta_path[:] = [f for f in sys.meta_path if f is not finder and type(f).__name__ != "_RegisterAfterImport"]
        sys.path.remove(REFERENCE)
        for k in [k for k in sys.modules if k == "nequip" or k.startswith("nequip.") or k.split(".")[0] in mrg._Finder.TOPS]:
            del sys.modules[k]


def _build_reference(ref):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return ref["model"].NequIPGNNModel(**HYPER)


def test_reference_builder_output_is_all_reference_modules(ref):
    model = _build_reference(ref)
    assert type(model).__module__ == "nequip.nn.graph_model"
    chain = model.model.func
    assert [n for n, _ in chain.named_children()] == [
        "type_embed", "spharm", "edge_norm", "bessel_encode", "factor", "layer0_convnet", "layer1_convnet",
        "layer2_convnet", "per_atom_energy_readout", "per_type_energy_scale_shift", "total_energy_sum"]
    assert all(type(m).__module__.startswith("nequip.") for m in chain.children())
    assert type(chain.layer1_convnet.conv).__module__ == "nequip.nn.interaction_block"


def test_full_modifier_through_the_reference_machinery(ref, monkeypatch):
    from nequip_amd.integrations import nequip_full
    from nequip_amd.model import NequIPGNNModel

    RefConvNetLayer = ref["nn"].ConvNetLayer
    nequip_full.register_full()  # default target: nequip.nn.convnetlayer.ConvNetLayer
    mmu = sys.modules["nequip.nn.model_modifier_utils"]
    fn = RefConvNetLayer.enable_NequipAMD_full
    assert mmu.is_model_modifier(fn) and mmu.is_persistent_model_modifier(fn) is False
    assert mmu.get_model_modifier_unsupported_devices(fn) == ["cpu"]

    model = _build_reference(ref)
    modifiers = ref["modify_utils"].get_all_modifiers(model)
    assert nequip_full.FULL_MODIFIER_NAME in modifiers and "enable_OpenEquivariance" in modifiers

    keys_before = list(model.state_dict().keys())
    values_before = {k: v.detach().clone() for k, v in model.state_dict().items()}
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.version, "hip", "7.0", raising=False)
    converted = ref["modify_utils"].modify(model, [{"modifier": nequip_full.FULL_MODIFIER_NAME}])

    # containers stay the reference's, everything on the path is the mirror
    assert type(converted).__module__ == "nequip.nn.graph_model"
    assert type(converted.model).__module__ == "nequip_amd.nn.grad_output"
    chain = converted.model.func
    assert type(chain).__module__ == "nequip.nn._graph_mixin"
    assert all(type(m).__module__.startswith("nequip_amd.") for m in chain.children())

    # state dict: same keys in the same order, same values (nequip issue #572: checkpoints load with or without modifiers)
    assert list(converted.state_dict().keys()) == keys_before
    for k, v in converted.state_dict().items():
        assert torch.equal(v, values_before[k]), k

    # module for module the natively built model (what bench.py times), same fusion plan
    native = NequIPGNNModel(**HYPER)
    nchain = native.model.func
    assert [n for n, _ in chain.named_children()] == [n for n, _ in nchain.named_children()]
    for (name, a), (_, b) in zip(chain.named_children(), nchain.named_children()):
        assert type(a) is type(b), name
        # (the reference chain also carries the model's input fields -- pos, edge_index, atom_types -- through its irreps)
        assert {k: str(a.irreps_out[k]) for k in b.irreps_out} == {k: str(v) for k, v in b.irreps_out.items()}, name
    for name in ("layer0_convnet", "layer1_convnet", "layer2_convnet"):
        a, b = getattr(chain, name), getattr(nchain, name)
        assert a.defer_gate == b.defer_gate is True
        assert a.resnet == b.resnet
        assert a.conv.use_sc == b.conv.use_sc and a.conv.is_first_layer == b.conv.is_first_layer
        assert a.conv.tp_scatter.tp.weight_numel == b.conv.tp_scatter.tp.weight_numel
        assert str(a.conv.tp_scatter.irreps_mid) == str(b.conv.tp_scatter.irreps_mid)
        assert a.conv.tp_scatter.instructions == b.conv.tp_scatter.instructions
        assert a.equivariant_nonlin._op_key == b.equivariant_nonlin._op_key  # same irreps, activations, normalize2mom constants
        assert torch.equal(a.conv.avg_num_neighbors_norm.norm_const, b.conv.avg_num_neighbors_norm.norm_const)
        assert a.conv.edge_mlp.dims == b.conv.edge_mlp.dims
    assert chain.bessel_encode.factor == nchain.bessel_encode.factor and chain.factor._folded
    assert chain.edge_norm.symmetric and float(chain.edge_norm._rmax_recip) == float(nchain.edge_norm._rmax_recip)
    assert "_scale_shift" in chain.per_atom_energy_readout.__dict__
    assert chain.per_atom_energy_readout.__dict__["_scale_shift"][0] is chain.per_type_energy_scale_shift
    assert torch.equal(chain.per_type_energy_scale_shift.scales, nchain.per_type_energy_scale_shift.scales)
    assert torch.equal(chain.per_type_energy_scale_shift.shifts, nchain.per_type_energy_scale_shift.shifts)

    # and the mirror loads the converted model's parameters one to one (what a GPU evaluation would run with)
    missing, unexpected = native.load_state_dict(converted.state_dict(), strict=False)
    assert not missing and all(k.endswith("_empty") for k in unexpected)


def test_full_modifier_leaves_other_modules_alone_and_reads_the_e3nn_layout(ref):
    from nequip_amd.integrations import nequip_full

    model = _build_reference(ref)
    gate = model.model.func.layer1_convnet.equivariant_nonlin
    assert hasattr(gate.act_scalars, "acts") and hasattr(gate.act_scalars.acts[0], "f")  # the e3nn shape, not a list
    kw = nequip_full._gate_kwargs(gate)
    assert kw == dict(nonlinearity_type="gate", nonlinearity_scalars={"e": "silu", "o": "tanh"},
                      nonlinearity_gates={"e": "silu", "o": "tanh"})
    shape = nequip_full._mlp_shape(model.model.func.layer1_convnet.conv.edge_mlp)
    assert shape == dict(depth=1, width=16, nonlinearity="silu", bias=False)
    assert nequip_full._avg_num_neighbors(model.model.func.layer1_convnet.conv.avg_num_neighbors_norm, ["H", "O"]) == pytest.approx(20.0)

    class Extra(torch.nn.Module):  # a module the table does not know is not touched
        def forward(self, data):
            return data

    model.model.func.add_module("extra", Extra())
    out = nequip_full.convert(model)
    assert type(out.model.func.extra) is Extra


OTHER_SHAPES = {
    # configs/tutorial.yaml's hyper-parameters (BASELINE cfg-1): parity irreps, four layers, radial MLP of depth 2 / width 64
    "tutorial": dict(seed=5, model_dtype="float32", type_names=["C", "H", "O"], r_max=5.0, num_layers=4, l_max=1, parity=True,
                     num_features=32, radial_mlp_depth=2, radial_mlp_width=64, avg_num_neighbors=13.0),
    # per-edge-type cutoffs (asymmetric: the reverse-edge pairing must be off), per-type neighbour counts, norm nonlinearity,
    # trainable Bessel roots, no self-connection, resnet update
    "options": dict(seed=7, model_dtype="float32", type_names=["H", "O"], r_max=4.0, num_layers=3, l_max=2, parity=False,
                    num_features=8, radial_mlp_width=16, avg_num_neighbors={"H": 11.0, "O": 23.0},
                    per_edge_type_cutoff={"H": {"H": 3.0, "O": 3.5}, "O": 4.0}, convnet_nonlinearity_type="norm",
                    bessel_trainable=True, convnet_sc=False, convnet_resnet=True,
                    per_type_energy_shifts={"H": 0.5, "O": -0.25}),
}


@pytest.mark.parametrize("shape", sorted(OTHER_SHAPES))
def test_full_modifier_other_model_shapes(ref, monkeypatch, shape):
    """The same check on the tutorial's hyper-parameters and on a model that uses the options outside the benchmarked
    configs: the converted chain equals the natively built one module for module, with the reference's parameters."""
    from nequip_amd.integrations import nequip_full
    from nequip_amd.model import NequIPGNNModel

    hyper = OTHER_SHAPES[shape]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model = ref["model"].NequIPGNNModel(**hyper)
        native = NequIPGNNModel(**hyper)
    keys_before = list(model.state_dict().keys())
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.version, "hip", "7.0", raising=False)
    nequip_full.register_full()
    converted = ref["modify_utils"].modify(model, [{"modifier": nequip_full.FULL_MODIFIER_NAME}])
    assert list(converted.state_dict().keys()) == keys_before
    chain, nchain = converted.model.func, native.model.func
    assert [n for n, _ in chain.named_children()] == [n for n, _ in nchain.named_children()]
    for (name, a), (_, b) in zip(chain.named_children(), nchain.named_children()):
        assert type(a) is type(b), name
        assert {k: str(a.irreps_out[k]) for k in b.irreps_out} == {k: str(v) for k, v in b.irreps_out.items()}, name
        if hasattr(b, "conv"):
            assert a.defer_gate == b.defer_gate and a.resnet == b.resnet, name
            assert type(a.equivariant_nonlin) is type(b.equivariant_nonlin), name
            assert a.conv.use_sc == b.conv.use_sc and a.conv.paired_radial_ok == b.conv.paired_radial_ok, name
            assert a.conv.edge_mlp.dims == b.conv.edge_mlp.dims, name
            assert a.conv.tp_scatter.instructions == b.conv.tp_scatter.instructions, name
            assert torch.equal(a.conv.avg_num_neighbors_norm.norm_const, b.conv.avg_num_neighbors_norm.norm_const), name
    assert chain.edge_norm.symmetric == nchain.edge_norm.symmetric
    assert torch.equal(chain.edge_norm._rmax_recip, nchain.edge_norm._rmax_recip)
    assert chain.bessel_encode.trainable == nchain.bessel_encode.trainable
    assert isinstance(chain.bessel_encode.bessel_weights, torch.nn.Parameter) == hyper.get("bessel_trainable", False)
    assert ("_fuse_radial" in chain.spharm.__dict__) == ("_fuse_radial" in nchain.spharm.__dict__)
    assert ("_scale_shift" in chain.per_atom_energy_readout.__dict__) == ("_scale_shift" in nchain.per_atom_energy_readout.__dict__)
    missing, unexpected = native.load_state_dict(converted.state_dict(), strict=False)
    assert not missing and all(k.endswith("_empty") for k in unexpected)
