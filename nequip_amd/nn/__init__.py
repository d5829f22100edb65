from ._graph_mixin import GraphModuleMixin, SequentialGraphNetwork  # noqa: F401
from ._ghost_exchange import (  # noqa: F401
    GhostExchangeModule,
    LAMMPSMLIAPGhostExchangeModule,
    NoOpGhostExchangeModule,
)
from ._tp_scatter_base import TensorProductScatter  # noqa: F401
from ._topology import EdgeTopology, topology_cache  # noqa: F401
from .atomwise import AtomwiseReduce, PerTypeScaleShift  # noqa: F401
from .convnetlayer import ConvNetLayer  # noqa: F401
from .grad_output import ForceStressOutput  # noqa: F401
from .graph_model import GraphModel  # noqa: F401
from .interaction_block import InteractionBlock  # noqa: F401
from .misc import ApplyFactor  # noqa: F401
from .mlp import ScalarMLP, ScalarMLPFunction  # noqa: F401
from .norm import AvgNumNeighborsNorm  # noqa: F401
from .utils import scatter, tp_path_exists, with_edge_vectors_  # noqa: F401
