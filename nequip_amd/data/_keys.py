"""Field names of the ``AtomicDataDict`` data model.

The *strings* are the reference's (``nequip/data/_keys.py:12-115``): data dicts built for nequip can be fed to these
modules and vice versa.  Only the fields the hot path reads or writes are listed, grouped by who produces them; each
group is one table ``CONSTANT_NAME -> field string`` that is exported into the module namespace below.
"""

from typing import Dict

# what the caller provides: geometry, species, graph (row 0 of edge_index = convolution centre / destination,
# row 1 = neighbour / source), frame bookkeeping of a batch
_INPUTS: Dict[str, str] = dict(
    POSITIONS_KEY="pos",
    ATOM_TYPE_KEY="atom_types",
    CELL_KEY="cell",
    PBC_KEY="pbc",
    EDGE_INDEX_KEY="edge_index",
    EDGE_CELL_SHIFT_KEY="edge_cell_shift",
    EDGE_TRANSPOSE_PERM_KEY="edge_transpose_perm",
    BATCH_KEY="batch",
    NUM_NODES_KEY="num_atoms",
    LMP_MLIAP_DATA_KEY="lmp_mliap_data",
    NUM_LOCAL_GHOST_NODES_KEY="num_local_ghost_atoms",
)

# per-edge quantities written by the embedding modules (an edge-vector based caller provides `edge_vectors` itself)
_EDGE_FIELDS: Dict[str, str] = dict(
    EDGE_VECTORS_KEY="edge_vectors",
    EDGE_LENGTH_KEY="edge_lengths",
    NORM_LENGTH_KEY="normed_edge_lengths",
    EDGE_TYPE_KEY="edge_type_flat",
    EDGE_CUTOFF_KEY="edge_cutoff",
    EDGE_ATTRS_KEY="edge_attrs",
    EDGE_EMBEDDING_KEY="edge_embedding",
)

# per-node quantities flowing through the convolution layers
_NODE_FIELDS: Dict[str, str] = dict(
    NODE_ATTRS_KEY="node_attrs",
    NODE_FEATURES_KEY="node_features",
    FEATURE_NORM_FACTOR_KEY="feature_norm_factor",
)

# model outputs
_OUTPUTS: Dict[str, str] = dict(
    PER_ATOM_ENERGY_KEY="atomic_energy",
    TOTAL_ENERGY_KEY="total_energy",
    FORCE_KEY="forces",
    EDGE_FORCE_KEY="edge_forces",
    STRESS_KEY="stress",
    VIRIAL_KEY="virial",
)

ALL_KEYS: Dict[str, str] = {**_INPUTS, **_EDGE_FIELDS, **_NODE_FIELDS, **_OUTPUTS}
assert len(set(ALL_KEYS.values())) == len(ALL_KEYS), "field strings must be unique"
globals().update(ALL_KEYS)

__all__ = sorted(ALL_KEYS)
