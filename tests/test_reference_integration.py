"""`enable_NequipAMD` against the REFERENCE's own modifier machinery (runs only where /root/reference exists, i.e. in the
build container; the GPU box has no reference and skips).

nequip is imported from /root/reference with the inert stand-ins of tests/golden/make_reference_golden.py for the
uninstalled e3nn / training stack.  What is exercised is nequip's real code: the `model_modifier` attribute protocol
(`nequip/nn/model_modifier_utils.py:22-89`), modifier discovery over a module tree
(`nequip/model/modify_utils.py:35-63`: classmethods found by `inspect.getmembers`, names must be unique) and
`replace_submodules` (`model_modifier_utils.py:92-107`) driven by the factory that `register()` installs on nequip's
`TensorProductScatter` (`nequip/nn/_tp_scatter_base.py`).  The e3nn `TensorProduct` inside the reference class is a
stand-in object here, so no reference arithmetic runs.
"""

import importlib.util
import os
import sys

import pytest
import torch

REFERENCE = os.environ.get("NEQUIP_REFERENCE", "/root/reference")
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REFERENCE, "nequip")), reason="reference tree not present")


@pytest.fixture(scope="module")
def ref():
    spec = importlib.util.spec_from_file_location(
        "make_reference_golden", os.path.join(os.path.dirname(__file__), "golden", "make_reference_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    added_finder = mod._Finder()
    sys.meta_path.insert(0, added_finder)
    sys.path.insert(0, REFERENCE)
    import warnings

    real_filterwarnings = warnings.filterwarnings

    def tolerant(action, message="", category=Warning, *a, **k):  # stand-in objects are not warning classes
        if isinstance(category, type):
            real_filterwarnings(action, message, category, *a, **k)

    warnings.filterwarnings = tolerant
    try:
        import nequip.model.modify_utils as modify_utils
        import nequip.nn._tp_scatter_base as tps
        import nequip.nn.model_modifier_utils as mmu

        yield dict(modify_utils=modify_utils, tps=tps, mmu=mmu)
    finally:
        warnings.filterwarnings = real_filterwarnings
        sys.meta_path.remove(added_finder)
        sys.path.remove(REFERENCE)
        for k in [k for k in sys.modules if k == "nequip" or k.startswith("nequip.") or k.split(".")[0] in mod._Finder.TOPS]:
            del sys.modules[k]


def test_register_follows_the_reference_modifier_protocol(ref):
    from nequip_amd.integrations import nequip_extension as ext

    RefTPS = ref["tps"].TensorProductScatter
    ext.register()  # default target: nequip.nn._tp_scatter_base.TensorProductScatter
    assert hasattr(RefTPS, "enable_NequipAMD")
    fn = RefTPS.enable_NequipAMD
    mmu = ref["mmu"]
    assert mmu.is_model_modifier(fn)
    assert mmu.is_persistent_model_modifier(fn) is False
    assert mmu.is_private_model_modifier(fn) is False
    assert mmu.get_model_modifier_unsupported_devices(fn) == ["cpu"]
    assert mmu.get_model_modifier_supported_compile_modes(fn) == ["aotinductor"]
    # the shape of the upstream adapters is the template: same decorator settings as enable_OpenEquivariance
    oeq = RefTPS.enable_OpenEquivariance
    assert mmu.is_persistent_model_modifier(oeq) == mmu.is_persistent_model_modifier(fn)
    assert mmu.get_model_modifier_unsupported_devices(oeq) == mmu.get_model_modifier_unsupported_devices(fn)


def test_reference_discovery_and_replacement(ref, monkeypatch):
    from nequip_amd.integrations import nequip_extension as ext

    RefTPS = ref["tps"].TensorProductScatter
    ext.register()

    class Layer(torch.nn.Module):
        def __init__(self):
            super().__init__()
            # the reference constructor; its e3nn TensorProduct is a stand-in object (never called here)
            self.tp_scatter = RefTPS.__new__(RefTPS)
            torch.nn.Module.__init__(self.tp_scatter)
            self.tp_scatter.feature_irreps_in = "4x0e+4x1o"
            self.tp_scatter.irreps_edge_attr = "1x0e+1x1o"
            self.tp_scatter.irreps_mid = "4x0e+4x1o+4x1o+4x0e"
            self.tp_scatter.instructions = [(0, 0, 0, "uvu", True), (0, 1, 1, "uvu", True), (1, 0, 2, "uvu", True), (1, 1, 3, "uvu", True)]
            self.tp_scatter.model_dtype = torch.float32
            self.tp_scatter.tp = torch.nn.Identity()
            self.tp_scatter.register_buffer("_dummy", torch.zeros(1))

    class Model(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.layers = torch.nn.ModuleList([Layer(), Layer()])

    model = Model()
    modifiers = ref["modify_utils"].get_all_modifiers(model)
    assert "enable_NequipAMD" in modifiers and "enable_OpenEquivariance" in modifiers

    # the factory itself: nequip's replace_submodules swaps every reference TensorProductScatter for ours, keeping `tp`
    from nequip_amd.nn import TensorProductScatter as HipTPS

    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.version, "hip", "7.0", raising=False)
    old_tp = model.layers[0].tp_scatter.tp
    # instruction tuples in the reference carry (i1, i2, i_out, mode, has_weight); ours accepts the same 5-tuples
    new_model = modifiers["enable_NequipAMD"](model)
    for layer in new_model.layers:
        assert isinstance(layer.tp_scatter, HipTPS)
    assert new_model.layers[0].tp_scatter.tp is old_tp


def test_entry_point_autoload_registers_on_import_of_the_reference_module():
    """The `nequip.extension` / `init_always` entry point target (pyproject.toml): loaded BEFORE nequip.nn exists, it must
    attach `enable_NequipAMD` as soon as `nequip.nn._tp_scatter_base` is imported by anybody."""
    spec = importlib.util.spec_from_file_location(
        "make_reference_golden", os.path.join(os.path.dirname(__file__), "golden", "make_reference_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    finder = mod._Finder()
    for k in [k for k in sys.modules if k == "nequip" or k.startswith("nequip.")]:
        del sys.modules[k]
    sys.modules.pop("nequip_amd.integrations.nequip_autoload", None)
    sys.meta_path.insert(0, finder)
    sys.path.insert(0, REFERENCE)
    import warnings

    real = warnings.filterwarnings
    warnings.filterwarnings = lambda action, message="", category=Warning, *a, **k: (
        real(action, message, category, *a, **k) if isinstance(category, type) else None)
    try:
        import nequip_amd.integrations.nequip_autoload as auto  # what `ep.load()` does, nequip.nn not imported yet

        assert "nequip.nn._tp_scatter_base" not in sys.modules
        assert any(isinstance(f, auto._RegisterAfterImport) for f in sys.meta_path)
        import nequip.nn._tp_scatter_base as tps

        assert hasattr(tps.TensorProductScatter, "enable_NequipAMD")
        assert not any(isinstance(f, auto._RegisterAfterImport) for f in sys.meta_path)  # one-shot
    finally:
        warnings.filterwarnings = real
        sys.meta_path[:] = [f for f in sys.meta_path if f is not finder and type(f).__name__ != "_RegisterAfterImport"]
        sys.path.remove(REFERENCE)
        for k in [k for k in sys.modules if k == "nequip" or k.startswith("nequip.") or k.split(".")[0] in mod._Finder.TOPS]:
            del sys.modules[k]


def test_pyproject_declares_the_entry_point():
    import tomli

    cfg = tomli.load(open(os.path.join(os.path.dirname(__file__), "..", "pyproject.toml"), "rb"))
    eps = cfg["project"]["entry-points"]["nequip.extension"]
    assert eps["init_always"] == "nequip_amd.integrations.nequip_autoload"
    importlib.import_module(eps["init_always"])  # importable without nequip installed (defers registration)
