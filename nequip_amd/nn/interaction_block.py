"""InteractionBlock: one message-passing convolution (mirror of ``nequip/nn/interaction_block.py:21-207``).

``linear_1 -> 1/sqrt(avg_num_neighbors) -> TensorProductScatter(edge_mlp(edge_embedding)) -> linear_2 (+ sc)``
with the instruction list, sorted ``irreps_mid`` and parameter names of the reference; the tensor product /
gather / scatter is the fused HIP ``TensorProductScatter``.
"""

import contextlib
import os
from typing import Dict, Optional, Sequence, Union

import torch

from ..data import AtomicDataDict
from ..o3 import _node_kernels
from ..o3.irreps import Irreps
from ..o3.modules import FullyConnectedTensorProduct, Linear
from ..utils.wgrad import differentiable_parameters
from . import _paired_radial, _radial_tp_ops, _segmented
from ._ghost_exchange import NoOpGhostExchangeModule
from ._graph_mixin import GraphModuleMixin
from ._topology import topology_cache
from ._tp_scatter_base import TensorProductScatter
from .mlp import ScalarMLPFunction
from .norm import AvgNumNeighborsNorm


def uvu_paths(features_in: Irreps, edge_attr: Irreps, features_out: Irreps):
    """The 'uvu' paths of one convolution and the irreps they produce (semantics of
    ``nequip/nn/interaction_block.py:89-109``): every (input block a, edge-attr block b) pair contributes one path per
    irrep of ``ir_a x ir_b`` that the layer's output wants; each path gets its own output slot with the input block's
    multiplicity.  The slots are then ordered like ``Irreps.sort()`` (so that ``linear_2`` can simplify them), while
    the paths -- and with them the columns of the radial MLP's output -- keep their creation order.

    Returns ``(irreps_mid sorted, [(a, b, slot, "uvu", True), ...])``."""
    created = []
    for a, (mul, ir_a) in enumerate(features_in):
        for b, (_, ir_b) in enumerate(edge_attr):
            created.extend((a, b, (mul, ir)) for ir in ir_a * ir_b if ir in features_out)
    mid_sorted, slot_of, _ = Irreps([blk for _, _, blk in created]).sort()
    return mid_sorted, [(a, b, slot_of[n], "uvu", True) for n, (a, b, _) in enumerate(created)]


from ..utils.tracing import traceable

class InteractionBlock(GraphModuleMixin, torch.nn.Module):
    use_sc: bool
    paired_radial_ok: bool = True  # the edge embedding depends on |r| only (see forward)

    def __init__(self, irreps_in, irreps_out, radial_mlp_depth: int = 1, radial_mlp_width: int = 8,
                 use_sc: bool = True, is_first_layer: bool = False, type_names: Optional[Sequence[str]] = None,
                 avg_num_neighbors: Optional[Union[float, Dict[str, float]]] = None) -> None:
        super().__init__()
        self._init_irreps(
            irreps_in=irreps_in,
            required_irreps_in=[
                AtomicDataDict.EDGE_EMBEDDING_KEY,
                AtomicDataDict.EDGE_ATTRS_KEY,
                AtomicDataDict.NODE_FEATURES_KEY,
                AtomicDataDict.NODE_ATTRS_KEY,
            ],
            irreps_out={AtomicDataDict.NODE_FEATURES_KEY: irreps_out},
        )
        self.avg_num_neighbors_norm = AvgNumNeighborsNorm(avg_num_neighbors=avg_num_neighbors, type_names=type_names)
        self.use_sc = use_sc
        feature_irreps_in = self.irreps_in[AtomicDataDict.NODE_FEATURES_KEY]
        feature_irreps_out = self.irreps_out[AtomicDataDict.NODE_FEATURES_KEY]
        irreps_edge_attr = self.irreps_in[AtomicDataDict.EDGE_ATTRS_KEY]

        self.linear_1 = Linear(irreps_in=feature_irreps_in, irreps_out=feature_irreps_in)

        irreps_mid, instructions = uvu_paths(feature_irreps_in, irreps_edge_attr, feature_irreps_out)

        self.tp_scatter = TensorProductScatter(feature_irreps_in, irreps_edge_attr, irreps_mid, instructions)
        # non-uniform multiplicities (S / M / L presets): the convolution as uniform pieces over channel ranges, each on the
        # structure-specialised kernels (nn/_segmented.py).  The pieces own no parameters and no persistent buffers.
        segs = _segmented.channel_segments(
            feature_irreps_in, irreps_edge_attr, irreps_mid, instructions,
            lambda a, b, c, d: TensorProductScatter(a, b, c, d))
        self._segments = segs
        if segs is not None:
            self.segment_tps = torch.nn.ModuleList([sg.tp for sg in segs])
            self.register_buffer("_segment_out_perm", _segmented.output_permutation(segs, Irreps(str(irreps_mid)).dim),
                                 persistent=False)

        self.edge_mlp = ScalarMLPFunction(
            input_dim=self.irreps_in[AtomicDataDict.EDGE_EMBEDDING_KEY].num_irreps,
            output_dim=self.tp_scatter.tp.weight_numel,
            hidden_layers_depth=radial_mlp_depth,
            hidden_layers_width=radial_mlp_width,
            nonlinearity="silu",
            bias=False,
            forward_weight_init=True,
        )

        if self._segments is not None:
            for sg in self._segments:
                sg.mlp = _segmented._SegmentMLP(self.edge_mlp, sg.w_cols) if radial_mlp_depth >= 1 else None

        self.linear_2 = Linear(irreps_in=irreps_mid.simplify(), irreps_out=feature_irreps_out)

        self.sc = None
        if self.use_sc:
            self.sc = FullyConnectedTensorProduct(
                feature_irreps_in, self.irreps_in[AtomicDataDict.NODE_ATTRS_KEY], feature_irreps_out
            )
        # ghost rows of a domain-decomposed caller are fetched here (no-op unless enable_LAMMPSMLIAPGhostExchange ran)
        self.ghost_exchange = NoOpGhostExchangeModule(
            field=AtomicDataDict.NODE_FEATURES_KEY, irreps_in={AtomicDataDict.NODE_FEATURES_KEY: feature_irreps_in}
        )
        self.is_first_layer = is_first_layer

    def _fused_node_stage(self, data, h: torch.Tensor, gate_meta, num_local_nodes: int, gate_key: str = ""):
        """``(x1, sc)`` of this block from the previous layer's pre-gate rows in one launch, or None when the conditions of
        the fused kernel do not hold (then the caller applies the gate and takes the separate launches)."""
        norm = self.avg_num_neighbors_norm
        if (self.sc is None or not h.is_cuda or h.dtype != torch.float32 or self.training
                or not _node_kernels.fusion_enabled() or not norm.norm_shortcut or norm.norm_key in data
                or self.sc._meta is None or self.linear_1.weight_numel == 0
                or differentiable_parameters(self.training, self.sc.weight)
                or differentiable_parameters(self.training, self.linear_1.weight)):
            return None
        table = data.get("_nqa_node_attrs_table")
        if table is None or table.shape[0] > 16:
            return None
        types = data[AtomicDataDict.ATOM_TYPE_KEY].view(-1)[: h.shape[0]].contiguous()
        if traceable():  # the same launch as a dispatcher-op pair (o3/_node_ops.py::node_stage)
            from ..o3 import _node_ops

            return _node_ops.node_stage(h, types, self.linear_1.traced_weights(), self.sc.traced_weights_typed(table),
                                        gate_key, self.linear_1._op_key, self.sc._op_key, float(norm.norm_scalar))
        wp1 = self.linear_1.eval_weights(h.device, h.dtype)
        wps = self.sc.eval_weights_typed(table, h.dtype)
        return _node_kernels.fused_node_stage(h, types, gate_meta, wp1, self.linear_1._meta, float(norm.norm_scalar), wps,
                                              self.sc._meta)

    def _use_segments(self, x: torch.Tensor, emb: torch.Tensor) -> bool:
        """Channel-segment evaluation: float32 on the GPU, not while tracing (the dispatcher-op form takes any irreps), every
        piece has a structure-specialised kernel and a radial-MLP view."""
        segs = self._segments
        if segs is None or not x.is_cuda or x.dtype != torch.float32 or emb.dtype != torch.float32 or traceable():
            return False
        ok = self.__dict__.get("_segments_ok")
        if ok is None:
            ok = all(sg.mlp is not None and sg.tp._get_kernels().has_spec(torch.float32) for sg in segs)
            self.__dict__["_segments_ok"] = ok
        return ok and os.environ.get("NQA_NO_SEGMENTS", "") in ("", "0")

    def _segmented_conv(self, data, x, emb, edge_index) -> torch.Tensor:
        attrs = data[AtomicDataDict.EDGE_ATTRS_KEY]
        topo = pairing = emb_half = queue = None
        outs = []
        for sg in self._segments:
            sg.to(x.device)
            mlp = sg.mlp.sync()
            x_s = x.index_select(1, sg.x_cols)
            if (self.paired_radial_ok and AtomicDataDict.POSITIONS_KEY in data
                    and _paired_radial.available(mlp, sg.tp, x_s, emb)):
                if topo is None:
                    topo = topology_cache.get(edge_index[0], edge_index[1], x.size(0))
                    pairing = topo.pairing(data.get(AtomicDataDict.EDGE_CELL_SHIFT_KEY))
                if pairing is not None:
                    if emb_half is None:
                        emb_half = data.get("_nqa_edge_embedding_pairs")
                        if emb_half is None or emb_half.shape[0] != pairing.num_pairs:
                            emb_half = _paired_radial.pair_rows(emb, pairing)
                            data["_nqa_edge_embedding_pairs"] = emb_half
                    if (queue is None and emb_half.requires_grad and _paired_radial.RadialBackwardQueue.enabled()
                            and not differentiable_parameters(self.training, self.edge_mlp.mlp[2].weight)):
                        queue = data.get("_nqa_radial_queue")
                        if queue is None:
                            queue = data["_nqa_radial_queue"] = _paired_radial.RadialBackwardQueue(x.device)
                    outs.append(_paired_radial.paired_radial_tp(mlp, sg.tp, emb, x_s, attrs, topo, pairing, emb_half,
                                                                queue))
                    continue
            outs.append(sg.tp(x=x_s, edge_attr=attrs, edge_weight=mlp(emb), edge_dst=edge_index[0],
                              edge_src=edge_index[1]))
        for sg in self._segments:
            sg.mlp.release()  # no graph-attached tensors stay on the module (deepcopy of a model in training)
        return torch.cat(outs, dim=1).index_select(1, self._segment_out_perm)

    def forward(self, data: AtomicDataDict.Type) -> AtomicDataDict.Type:
        if AtomicDataDict.LMP_MLIAP_DATA_KEY in data:
            num_local_nodes = int(data[AtomicDataDict.LMP_MLIAP_DATA_KEY].nlocal)
        else:
            num_local_nodes = AtomicDataDict.num_nodes(data)
        x = data[AtomicDataDict.NODE_FEATURES_KEY]
        # (`[:num_local_nodes]` as in the reference, interaction_block.py:166-168,199 -- only when there are ghost rows:
        # a no-op slice still records a SliceBackward whose backward is a zero fill + copy of the whole gradient)
        if not self.is_first_layer and x.shape[0] != num_local_nodes:
            x = x[:num_local_nodes]

        # The previous layer left its Gate to this block (ConvNetLayer.defer_gate): `x` holds the PRE-gate rows.  Gate,
        # linear_1 (with 1/sqrt(avg_num_neighbors)) and the typed self-connection then run as ONE launch, their backward
        # + the gate's backward as one more (o3/_node_kernels.py::fused_node_stage); otherwise the gate is applied here.
        pregate = data.pop("_nqa_pregate", None)
        fused = None
        if pregate is not None:
            gate_meta = pregate[1]
            fused = self._fused_node_stage(data, x, gate_meta, num_local_nodes, pregate[2])
            if fused is None:
                x = _node_kernels.apply_deferred_gate(pregate)

        sc = None
        sc_stream = None
        if fused is not None:
            x1, sc = fused
        elif self.sc is not None:
            node_attrs = data[AtomicDataDict.NODE_ATTRS_KEY]
            if not self.is_first_layer and node_attrs.shape[0] != num_local_nodes:
                node_attrs = node_attrs[:num_local_nodes]
            table = data.get("_nqa_node_attrs_table")
            # eval mode on the GPU: the self-connection is independent of linear_1 / the tensor product until `+ sc`
            # after linear_2, so it runs as a parallel branch on a side stream (its backward too: autograd runs a node's
            # backward on the stream of its forward); both kernels are small and latency-bound at 10k atoms
            if (not differentiable_parameters(self.training, self.sc.weight) and x.is_cuda
                    and _paired_radial.RadialBackwardQueue.enabled() and not traceable()):
                sc_stream = _paired_radial.side_stream(x.device, 1)
                sc_stream.wait_stream(torch.cuda.current_stream(x.device))
                x.record_stream(sc_stream)
            # (no side stream -- always the case while a compiler traces -- means no stream context at all: Dynamo
            # rejects `torch.cuda.stream(None)`)
            with (torch.cuda.stream(sc_stream) if sc_stream is not None else contextlib.nullcontext()):
                if table is not None and table.shape[0] <= 16:
                    sc = self.sc.forward_typed(x, data[AtomicDataDict.ATOM_TYPE_KEY].view(-1)[: x.shape[0]], table)
                else:
                    sc = self.sc(x, node_attrs)

        norm = self.avg_num_neighbors_norm
        if fused is not None:
            x = x1
        elif norm.norm_shortcut and x.is_cuda and norm.norm_key not in data:
            # one avg_num_neighbors for all types: 1/sqrt(avg) rides on the linear_1 launch (no separate N x D pass)
            x = self.linear_1(x, scale=norm.norm_scalar)
        else:
            x = self.linear_1(x)
            data[AtomicDataDict.NODE_FEATURES_KEY] = x
            data = norm(data)
            x = data[AtomicDataDict.NODE_FEATURES_KEY]

        # ghost exchange (interaction_block.py:184-190): later layers hold local rows only, the tensor product gathers from
        # ghosts too.  (The first layer's input is the type embedding, which already covers the ghosts.)
        if not self.is_first_layer and not isinstance(self.ghost_exchange, NoOpGhostExchangeModule):
            data[AtomicDataDict.NODE_FEATURES_KEY] = x
            data = self.ghost_exchange(data, ghost_included=False)
            x = data[AtomicDataDict.NODE_FEATURES_KEY]

        emb = data[AtomicDataDict.EDGE_EMBEDDING_KEY]
        edge_index = data[AtomicDataDict.EDGE_INDEX_KEY]
        if self._use_segments(x, emb):
            x = self._segmented_conv(data, x, emb, edge_index)
            if x.shape[0] != num_local_nodes:
                x = x[:num_local_nodes]
            if sc_stream is not None:
                cur = torch.cuda.current_stream(x.device)
                cur.wait_stream(sc_stream)
                sc.record_stream(cur)
            data[AtomicDataDict.NODE_FEATURES_KEY] = self.linear_2(x, addend=sc if self.sc is not None else None)
            return data
        pairing = None
        # Pairing evaluates the radial MLP once per (i <- j) / (j <- i) pair: valid because this model's edge embedding is
        # a function of the edge length alone (`paired_radial_ok`; a builder with per-edge-type or otherwise asymmetric
        # embeddings must set it to False).  Not used when the caller differentiates w.r.t. given edge vectors (LAMMPS
        # ML-IAP branch, no positions): the pair's radial gradient would be attributed to one of its two edges, and the
        # per-edge EDGE_FORCE values, unlike their per-atom sums, would differ from the reference's.
        if (self.paired_radial_ok and AtomicDataDict.POSITIONS_KEY in data
                and _paired_radial.available(self.edge_mlp, self.tp_scatter, x, emb)):
            # inference: the radial MLP depends on the edge length only, and the list holds both directions of every
            # interaction -- evaluate it once per pair (nn/_paired_radial.py); None if the list does not pair up
            topo = topology_cache.get(edge_index[0], edge_index[1], x.size(0))
            pairing = topo.pairing(data.get(AtomicDataDict.EDGE_CELL_SHIFT_KEY))
        if (pairing is None and traceable() and self.paired_radial_ok and AtomicDataDict.POSITIONS_KEY in data
                and _radial_tp_ops.usable(self.edge_mlp, self.tp_scatter, x, emb)):
            # while a tracer follows the model: radial MLP + tensor product as ONE dispatcher-op pair whose implementation
            # takes the pairing decision at run time (nn/_radial_tp_ops.py)
            x = _radial_tp_ops.radial_tp(self.edge_mlp, self.tp_scatter, emb, x, data[AtomicDataDict.EDGE_ATTRS_KEY],
                                         edge_index[0], edge_index[1], data.get(AtomicDataDict.EDGE_CELL_SHIFT_KEY))
        elif pairing is not None:
            # the representative rows of the embedding are the same for every layer: gathered once per evaluation
            emb_half = data.get("_nqa_edge_embedding_pairs")
            if emb_half is None or emb_half.shape[0] != pairing.num_pairs:
                emb_half = _paired_radial.pair_rows(emb, pairing)
                data["_nqa_edge_embedding_pairs"] = emb_half
            # the layers of one evaluation share a side-stream queue for their radial-MLP backward launches (eval mode,
            # gradient w.r.t. the embedding requested): see RadialBackwardQueue
            queue = None
            if (emb_half.requires_grad and _paired_radial.RadialBackwardQueue.enabled()
                    and not differentiable_parameters(self.training, self.edge_mlp.mlp[2].weight)):
                queue = data.get("_nqa_radial_queue")
                if queue is None:
                    queue = data["_nqa_radial_queue"] = _paired_radial.RadialBackwardQueue(x.device)
            x = _paired_radial.paired_radial_tp(
                self.edge_mlp, self.tp_scatter, emb, x, data[AtomicDataDict.EDGE_ATTRS_KEY], topo, pairing, emb_half,
                queue,
            )
        else:
            x = self.tp_scatter(
                x=x,
                edge_attr=data[AtomicDataDict.EDGE_ATTRS_KEY],
                edge_weight=self.edge_mlp(emb),
                edge_dst=edge_index[0],
                edge_src=edge_index[1],
            )
        if x.shape[0] != num_local_nodes:
            x = x[:num_local_nodes]

        if sc_stream is not None:
            cur = torch.cuda.current_stream(x.device)
            cur.wait_stream(sc_stream)
            sc.record_stream(cur)
        # linear_2 with the residual `+ sc` fused into the same launch
        x = self.linear_2(x, addend=sc if self.sc is not None else None)
        data[AtomicDataDict.NODE_FEATURES_KEY] = x
        return data
