"""AOTInductor packaging of a model whose kernels are ``torch.ops.nequip_amd.*`` dispatcher ops.

Mirror of the reference's compile path for the ``aotinductor`` mode -- ``nequip/scripts/compile.py:248-344``
(``nequip-compile``), ``nequip/utils/aot.py::aot_export_model`` and the loader
``nequip/model/inference_models/aotinductor.py:57-125``:

    make_fx (symbolic) trace of energy + forces (autograd inside the model)  ->  torch.export with dynamic
    num_nodes / num_edges  ->  torch._inductor.aoti_compile_and_package  ->  <name>.nequip.pt2

with the same metadata keys (``nequip_aoti_inputs`` / ``nequip_aoti_outputs``, ``nequip_custom_ops_libs`` +
the ``nequip_custom_ops_libs.txt`` zip entry, ``nequip/utils/aoti_metadata.py:5-54``).  The custom-ops entry names
``nequip_amd``: importing the package registers every ``torch.ops.nequip_amd.*`` op (schema, fake kernel, HIP
implementation through the C ABI), which is what the reference's ``import_custom_ops_libs`` does before
``aoti_load_package`` for its OpenEquivariance / cuEquivariance adapters.

The weight-only part of the traced graph (path normalisation folded into ``o3.Linear`` weights, the per-type contraction of
the self-connection weights, ...) is evaluated at export time (``utils/tracing.py::fold_constants``), so the package's
constants are what the ops consume and their packed / transposed / split images are built once per loaded package
(``utils/constcache.py``; the C++ registrations keep the same cache).

Runtimes: Python hosts (ASE calculator, torch-sim, scripts) import ``nequip_amd``; hosts without an interpreter (a LAMMPS pair
style, a C++ server) link ``libnequip_amd_torch.so`` (``csrc/torch_ops``, ``include/nequip_amd_torch.h``), which registers the
same inference ops from C++ -- ``nequip_amd_aoti_run`` is such a host.
"""

from __future__ import annotations

import importlib
import zipfile
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from ..data import AtomicDataDict
from .tracing import trace_model

NEQUIP_AOTI_INPUTS_KEY = "nequip_aoti_inputs"
NEQUIP_AOTI_OUTPUTS_KEY = "nequip_aoti_outputs"
NEQUIP_CUSTOM_OPS_LIBS_KEY = "nequip_custom_ops_libs"
_CUSTOM_OPS_LIBS_ENTRY = "nequip_custom_ops_libs.txt"
_AOT_METADATA_KEY = "aot_inductor.metadata"
AOTI_DEVICE_KEY = "AOTI_DEVICE_KEY"

# reference defaults for the ASE target (nequip/scripts/_compile_utils.py: ASE inputs / outputs)
ASE_INPUTS = [AtomicDataDict.POSITIONS_KEY, AtomicDataDict.EDGE_INDEX_KEY, AtomicDataDict.ATOM_TYPE_KEY,
              AtomicDataDict.CELL_KEY, AtomicDataDict.EDGE_CELL_SHIFT_KEY]
ASE_OUTPUTS = [AtomicDataDict.TOTAL_ENERGY_KEY, AtomicDataDict.PER_ATOM_ENERGY_KEY, AtomicDataDict.FORCE_KEY,
               AtomicDataDict.VIRIAL_KEY, AtomicDataDict.STRESS_KEY]


def embed_custom_ops_libs(pt2_path: str, libs: Sequence[str]) -> None:
    if libs:
        with zipfile.ZipFile(pt2_path, "a") as zf:
            zf.writestr(_CUSTOM_OPS_LIBS_ENTRY, " ".join(sorted(set(libs))))


def import_custom_ops_libs(pt2_path: str) -> None:
    with zipfile.ZipFile(pt2_path, "r") as zf:
        if _CUSTOM_OPS_LIBS_ENTRY not in zf.namelist():
            return
        for lib in zf.read(_CUSTOM_OPS_LIBS_ENTRY).decode().split():
            importlib.import_module(lib)


class _ListIO(torch.nn.Module):
    """Positional tensors in / out around the traced graph, weights as buffers (what torch.export wants)."""

    def __init__(self, gm, params, buffers, input_fields, output_fields):
        super().__init__()
        self.gm = gm
        self.input_fields, self.output_fields = list(input_fields), list(output_fields)
        self._pn, self._bn = list(params), list(buffers)
        for i, k in enumerate(self._pn):
            self.register_buffer(f"p{i}", params[k].detach().clone())
        for i, k in enumerate(self._bn):
            self.register_buffer(f"b{i}", buffers[k].detach().clone())

    def forward(self, *tensors):
        p = {k: getattr(self, f"p{i}") for i, k in enumerate(self._pn)}
        b = {k: getattr(self, f"b{i}") for i, k in enumerate(self._bn)}
        out = self.gm(p, b, dict(zip(self.input_fields, tensors)))
        return tuple(out[k] for k in self.output_fields)


def _field_dims(field: str, batch_map) -> Optional[Dict[int, object]]:
    if field in (AtomicDataDict.POSITIONS_KEY, AtomicDataDict.ATOM_TYPE_KEY, AtomicDataDict.BATCH_KEY):
        return {0: batch_map["node"]}
    if field == AtomicDataDict.EDGE_INDEX_KEY:
        return {1: batch_map["edge"]}
    if field in (AtomicDataDict.EDGE_CELL_SHIFT_KEY, AtomicDataDict.EDGE_VECTORS_KEY):
        return {0: batch_map["edge"]}
    if field in (AtomicDataDict.CELL_KEY, AtomicDataDict.NUM_NODES_KEY):
        g = batch_map.get("graph", torch.export.Dim.STATIC)
        return None if g is torch.export.Dim.STATIC else {0: g}
    return None


def aot_export_model(model: torch.nn.Module, data: AtomicDataDict.Type, output_path: str,
                     input_fields: Sequence[str] = tuple(ASE_INPUTS), output_fields: Sequence[str] = tuple(ASE_OUTPUTS),
                     batch_map: Optional[dict] = None, metadata: Optional[dict] = None,
                     inductor_configs: Optional[dict] = None, fold_constants: bool = True) -> str:
    """Trace, export and package ``model`` (eval mode, on the GPU) for the example ``data``; returns ``output_path``.
    ``batch_map``: ``{"graph" | "node" | "edge": torch.export.Dim}`` (defaults: one frame, dynamic nodes / edges).
    ``fold_constants``: evaluate the weight-only part of the graph now (``utils/tracing.py::fold_constants``)."""
    if not str(output_path).endswith(".nequip.pt2"):
        raise ValueError("AOTInductor packages are named `<name>.nequip.pt2` (nequip/scripts/compile.py:97-104)")
    model = model.eval()
    for name, mod in model.named_modules():  # no dispatcher-op form: say so before tracing starts
        if getattr(mod, "_per_edge_type", False):
            raise NotImplementedError(f"aot_export_model: `{name}` uses per_edge_type_cutoff, which has no traceable "
                                      "(compile) form; export a model with a single cutoff or run it eagerly")
    inputs = {k: data[k] for k in input_fields}
    device = inputs[AtomicDataDict.POSITIONS_KEY].device
    if device.type != "cuda":
        raise RuntimeError("aot_export_model: the kernels are GPU-only, the example data must live on the device")
    gm, params, buffers = trace_model(model, inputs, tracing_mode="symbolic", fold=fold_constants)
    for nd in list(gm.graph.nodes):  # unused lifted constants trip torch.export's lift_constants_pass
        if nd.op == "get_attr" and len(nd.users) == 0:
            gm.graph.erase_node(nd)
    gm.graph.eliminate_dead_code()
    gm.recompile()
    wrapped = _ListIO(gm, params, buffers, input_fields, output_fields)
    if batch_map is None:
        batch_map = {"graph": torch.export.Dim.STATIC,
                     "node": torch.export.Dim("num_nodes", min=2, max=1 << 26),
                     "edge": torch.export.Dim("num_edges", min=2, max=1 << 30)}
    args = tuple(inputs[k] for k in input_fields)
    dyn = tuple(_field_dims(k, batch_map) for k in input_fields)
    ep = torch.export.export(wrapped, args, dynamic_shapes=(dyn,), strict=False)
    md = dict(metadata or {})
    graph_md = getattr(model, "metadata", None)
    if isinstance(graph_md, dict):
        md = {**graph_md, **md}
    md[NEQUIP_AOTI_INPUTS_KEY] = " ".join(input_fields)
    md[NEQUIP_AOTI_OUTPUTS_KEY] = " ".join(output_fields)
    md[NEQUIP_CUSTOM_OPS_LIBS_KEY] = " ".join(sorted(set(md.get(NEQUIP_CUSTOM_OPS_LIBS_KEY, "").split()) | {"nequip_amd"}))
    md[AOTI_DEVICE_KEY] = str(device.type)
    cfg = dict(inductor_configs or {})
    cfg[_AOT_METADATA_KEY] = {k: str(v) for k, v in md.items()}
    path = torch._inductor.aoti_compile_and_package(ep, package_path=str(output_path), inductor_configs=cfg)
    embed_custom_ops_libs(path, md[NEQUIP_CUSTOM_OPS_LIBS_KEY].split())
    return path


class DictInputOutputWrapper(torch.nn.Module):
    """``AtomicDataDict`` in / out around the positional compiled callable (nequip/model/inference_models/utils.py)."""

    def __init__(self, compiled, input_keys: List[str], output_keys: List[str]):
        super().__init__()
        self.compiled, self.input_keys, self.output_keys = compiled, list(input_keys), list(output_keys)

    def forward(self, data: AtomicDataDict.Type) -> AtomicDataDict.Type:
        outs = self.compiled(*[data[k] for k in self.input_keys])
        res = dict(data)
        res.update(zip(self.output_keys, outs))
        return res


def load_aotinductor_model(compile_path: str, device="cuda", input_keys: Optional[List[str]] = None,
                           output_keys: Optional[List[str]] = None) -> Tuple[torch.nn.Module, dict]:
    """(model taking / returning AtomicDataDicts, metadata) from a ``.nequip.pt2`` package
    (nequip/model/inference_models/aotinductor.py:57-125)."""
    import_custom_ops_libs(str(compile_path))  # registers torch.ops.nequip_amd.* before the package loader runs
    compiled = torch._inductor.aoti_load_package(str(compile_path))
    metadata = dict(compiled.get_metadata())
    input_keys = input_keys or metadata[NEQUIP_AOTI_INPUTS_KEY].split()
    output_keys = output_keys or metadata[NEQUIP_AOTI_OUTPUTS_KEY].split()
    compile_device = metadata.get(AOTI_DEVICE_KEY, "cuda")
    if torch.device(compile_device).type != torch.device(device).type:
        raise RuntimeError(f"`{compile_path}` was compiled for `{compile_device}` and won't work with device={device}")
    return DictInputOutputWrapper(compiled, input_keys, output_keys), metadata
