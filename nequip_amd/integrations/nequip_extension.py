"""Drop-in registration with an installed ``nequip`` (the reference) -- the same seam OpenEquivariance and
cuEquivariance use.

The reference swaps accelerated kernels in through ``@model_modifier`` classmethods on
``nequip.nn._tp_scatter_base.TensorProductScatter`` (``enable_OpenEquivariance`` / ``enable_CuEquivariance``,
``nequip/nn/_tp_scatter_base.py:40-109``), discovered by ``nequip.model.modify`` via ``inspect.getmembers``
(``nequip/model/modify_utils.py:35-63``) and applied from configs, ``nequip-compile --modifiers`` or the LAMMPS
wrapper.  ``register()`` attaches ``enable_NequipAMD`` to that class in exactly the same form, so that

    model = nequip.model.modify(model, [{"modifier": "enable_NequipAMD"}])

replaces every ``TensorProductScatter`` of a built / loaded nequip model by the HIP-backed
``nequip_amd.nn.TensorProductScatter`` (same constructor arguments, ``new.tp = old.tp`` to keep the e3nn buffers
and therefore the state-dict keys, ``nequip/nn/_tp_scatter_base.py:71-74``).  It is meant to be called from a
``nequip.extension`` / ``init_always`` entry point (``nequip/__init__.py:23-38``) or explicitly by the user.

nequip and e3nn are not installed in the build container, so this module is exercised only by its unit test with
a stand-in class; see INTEGRATION.md for the binding a maintainer would add upstream.
"""

from __future__ import annotations

import torch

from ..nn import TensorProductScatter as HipTensorProductScatter
from ..nn.model_modifier_utils import model_modifier, replace_submodules

MODIFIER_NAME = "enable_NequipAMD"


def make_modifier(base_cls):
    """Build the ``enable_NequipAMD`` classmethod for ``base_cls`` (nequip's ``TensorProductScatter``)."""

    def enable_NequipAMD(cls, model):
        """Enable the MI355X-native (gfx950 HIP) fused tensor-product/scatter kernels of ``nequip_amd``."""
        if not torch.cuda.is_available() or torch.version.hip is None:
            raise RuntimeError("enable_NequipAMD requires a ROCm build of PyTorch and an AMD GPU")

        # compiled graph models (nequip-compile / the LAMMPS wrapper set this flag, nequip/nn/_tp_scatter_base.py:60,69)
        # get the dispatcher-op form of the kernels, which make_fx / torch.compile can trace
        use_ops = bool(getattr(model, "is_compile_graph_model", False))

        def factory(old):
            prev = torch.get_default_dtype()
            torch.set_default_dtype(old.model_dtype)
            try:
                new = HipTensorProductScatter(
                    feature_irreps_in=old.feature_irreps_in,
                    irreps_edge_attr=old.irreps_edge_attr,
                    irreps_mid=old.irreps_mid,
                    instructions=old.instructions,
                    use_dispatcher_ops=use_ops,
                )
            finally:
                torch.set_default_dtype(prev)
            # reuse old.tp to preserve e3nn's persistent buffers -> identical state-dict keys with or without
            # the modifier (https://github.com/mir-group/nequip/issues/572)
            new.tp = old.tp
            device = next((b.device for b in old.buffers()), None)
            if device is not None:
                new = new.to(device)
            return new

        return replace_submodules(model, cls, factory)

    return model_modifier(
        persistent=False,
        private=False,
        unsupported_devices=["cpu"],
        # `nequip-compile --mode aotinductor --modifiers enable_NequipAMD`: the module then runs as torch.ops.nequip_amd.*
        # dispatcher ops, which make_fx / torch.export / AOTInductor carry as extern calls; the package records
        # `nequip_amd` in its custom-ops entry (`_nequip_custom_ops_libs`), imported by the reference's loader before
        # `aoti_load_package` (nequip/utils/aoti_metadata.py:40-54).  Exercised end to end on the GPU by
        # tests/test_aot_inductor.py with this package's own mirror of that flow.  (No TorchScript form.)
        supported_compile_modes=["aotinductor"],
    )(classmethod(enable_NequipAMD))


def register(base_cls=None):
    """Attach ``enable_NequipAMD`` to nequip's ``TensorProductScatter`` (or to ``base_cls`` for tests)."""
    if base_cls is None:
        from nequip.nn._tp_scatter_base import TensorProductScatter as base_cls  # type: ignore
    if not hasattr(base_cls, MODIFIER_NAME):
        setattr(base_cls, MODIFIER_NAME, make_modifier(base_cls))
    return base_cls
