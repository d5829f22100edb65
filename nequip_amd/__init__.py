"""MI355X-native NequIP message-passing hot path.

Importing the package registers the ``torch.ops.nequip_amd.*`` dispatcher ops (schemas, fake kernels, autograd
formulas, HIP implementations through the C ABI): compiled artefacts name ``nequip_amd`` in their custom-ops entry and
the loader imports it before the package loader runs (``nequip/utils/aoti_metadata.py:40-54``)."""


def register_ops() -> None:
    """Idempotent: import the modules that define the dispatcher ops."""
    from .nn import _edge_vector_ops, _energy_head, _force_ops, _mlp_ops, _radial_tp_ops, _tp_scatter_ops  # noqa: F401
    from .nn.embedding import _edge_ops  # noqa: F401
    from .o3 import _node_ops  # noqa: F401


register_ops()
