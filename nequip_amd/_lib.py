"""ctypes binding of ``libnequip_amd.so`` -- the C ABI declared in ``include/nequip_amd.h``.

There is deliberately no fallback: if the HIP library has not been built (``python -m
nequip_amd.csrc.build`` / ``__graft_entry__.build()``) every kernel-backed op raises.  Nothing in this
package computes the hot path on the CPU.
"""

from __future__ import annotations

import ctypes
import os
import threading
from ctypes import POINTER, c_char_p, c_double, c_int32, c_int64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libnequip_amd.so")

NQA_OK = 0
NQA_F32 = 0
NQA_F64 = 1
NQA_MLP_FP32 = 0  # radial MLP GEMM on exact-fp32 MFMA
NQA_MLP_BF16X6 = 1  # ... on bf16 MFMA with 3-way split operands (fp32-accurate)
NQA_MLP_F16X3 = 2  # forward: scaled operands split into two fp16 terms, three products (fp32-accurate)
NQA_MLP_HINT_DEVICE_IS_IDLE = 0x100  # OR-ed into `mode` of nqa_radial_mlp_bwd: nothing runs next to this launch
NQA_LAYOUT_MUL_IR = 0
NQA_LAYOUT_IR_MUL = 1

NQA_PLAN_DIM_IN1 = 0
NQA_PLAN_DIM_IN2 = 1
NQA_PLAN_DIM_OUT = 2
NQA_PLAN_WEIGHT_NUMEL = 3
NQA_PLAN_NUM_INSTR = 4
NQA_PLAN_OUT_NEEDS_ZERO = 5
NQA_PLAN_YPART_WIDTH = 6
NQA_PLAN_HAS_SPECIALIZED = 7
NQA_PLAN_FUSED_ROWS_OK = 8

_P32 = POINTER(c_int32)


class GateBlock(ctypes.Structure):
    """``nqa_gate_block`` (include/nequip_amd.h): one block of a gate's output."""

    _fields_ = [("out_off", c_int32), ("d", c_int32), ("mul", c_int32), ("val_off", c_int32), ("gate_off", c_int32),
                ("act", c_int32), ("cst", c_double)]


class NodePart(ctypes.Structure):
    """``nqa_node_part`` (include/nequip_amd.h): one operand set + destination of ``nqa_node_fused``."""

    _fields_ = [("x", c_void_p), ("packed", c_void_p), ("chunk_table", c_void_p), ("instr_table", c_void_p),
                ("n_chunks", c_int32), ("n_instr", c_int32), ("n_types", c_int32), ("dim_in", c_int32),
                ("out", c_void_p), ("addend", c_void_p), ("dim_out", c_int32), ("accumulate", c_int32),
                ("scale", c_double), ("in_gate", c_void_p), ("n_in_gate", c_int32), ("pad", c_int32)]


# name -> (restype, argtypes); must list every symbol include/nequip_amd.h declares
SIGNATURES = {
    "nqa_abi_version": (c_int32, []),
    "nqa_last_error": (c_char_p, []),
    "nqa_lmax": (c_int32, []),
    "nqa_sh_lmax": (c_int32, []),
    "nqa_plan_create": (
        c_int32,
        [c_int32, _P32, _P32, _P32, c_int32, _P32, _P32, _P32, c_int32, _P32, _P32, _P32]
        + [c_int32, _P32, _P32, _P32, POINTER(c_double), c_int32, c_int32, POINTER(c_void_p)],
    ),
    "nqa_plan_destroy": (None, [c_void_p]),
    "nqa_plan_query": (c_int64, [c_void_p, c_int32]),
    "nqa_plan_image_bytes": (c_int64, [c_void_p]),
    "nqa_plan_image_write": (c_int32, [c_void_p, c_void_p, c_int64]),
    "nqa_csr_workspace_bytes": (c_int64, [c_int64, c_int64]),
    "nqa_csr_build": (
        c_int32,
        [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p],
    ),
    "nqa_tp_scatter_fwd": (
        c_int32,
        [c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]
        + [c_int64, c_int64, c_void_p],
    ),
    "nqa_tp_bwd_edge_workspace_bytes": (c_int64, [c_void_p, c_int32, c_int64]),
    "nqa_tp_scatter_bwd_edge": (
        c_int32,
        [c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]
        + [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p],
    ),
    "nqa_tp_bwd_fused_workspace_bytes": (c_int64, [c_void_p, c_int32, c_int64]),
    "nqa_tp_scatter_bwd_fused": (
        c_int32,
        [c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p]  # plan, image, dtype, x, y, w, grad_out
        + [c_void_p] * 5  # rowptr_dst, edge_id_dst, src_sorted, rowptr_src, edge_id_src
        + [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p],
    ),
    "nqa_tp_scatter_bwd_x": (
        c_int32,
        [c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]
        + [c_int64, c_int64, c_void_p],
    ),
    "nqa_frame_sum": (c_int32, [c_void_p, c_void_p, c_int64, c_int32, c_int64, c_void_p, c_void_p]),
    "nqa_edge_pairs_workspace_bytes": (c_int64, [c_int64]),
    "nqa_edge_pairs": (
        c_int32,
        [c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_void_p,
         c_void_p, c_void_p, c_void_p, c_void_p],
    ),
    "nqa_csr_from_pairs": (c_int32, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "nqa_pair_owner_workspace_bytes": (c_int64, [c_int64, c_int64]),
    "nqa_pair_owner_lists": (
        c_int32,
        [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_void_p,
         c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    ),
    "nqa_pair_owner_lists_guard": (c_int32, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    "nqa_pair_gather": (c_int32, [c_void_p, c_void_p, c_int64, c_int32, c_void_p, c_void_p]),
    "nqa_pair_expand": (c_int32, [c_void_p, c_void_p, c_int64, c_int64, c_int32, c_void_p, c_void_p]),
    "nqa_tp_scatter_fwd_paired": (
        c_int32,
        [c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]
        + [c_int64, c_int64, c_void_p, c_int64, c_void_p],
    ),
    "nqa_tp_scatter_bwd_edge_paired": (
        c_int32,
        [c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]
        + [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int64, c_void_p],
    ),
    "nqa_tp_scatter_bwd_fused_paired": (
        c_int32,
        [c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p]
        + [c_void_p] * 5
        + [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int64, c_void_p],
    ),
    "nqa_tp_bwd_pairs_workspace_bytes": (c_int64, [c_void_p, c_int32, c_int64]),
    "nqa_tp_scatter_bwd_pairs": (
        c_int32,
        [c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p]  # plan, image, dtype, x, y, w, grad_out
        + [c_void_p] * 7  # owner_rowptr, pair_other, pair_row, pair_edge_in, pair_edge_out, other_rowptr, other_slot
        + [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p],
    ),
    "nqa_tp_fwd_jvp_supported": (c_int32, [c_void_p, c_int32]),
    "nqa_tp_scatter_fwd_jvp": (
        c_int32,
        [c_void_p, c_void_p, c_int32] + [c_void_p] * 6  # plan, image, dtype, x, y, w, x_cot, y_cot, w_cot
        + [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_void_p],
    ),
    "nqa_tp_scatter_bwd_x_dual": (
        c_int32,
        [c_void_p, c_void_p, c_int32] + [c_void_p] * 5  # plan, image, dtype, y, w, y_cot, w_cot, grad_out
        + [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_void_p],
    ),
    "nqa_tp_bwd_pairs_dual_supported": (c_int32, [c_void_p, c_int32]),
    "nqa_tp_scatter_bwd_pairs_dual": (
        c_int32,
        [c_void_p, c_void_p, c_int32] + [c_void_p] * 7  # plan, image, dtype, x, x_cot, y, y_cot, w, w_cot, grad_out
        + [c_void_p] * 5  # owner_rowptr, pair_other, pair_row, pair_edge_in, pair_edge_out
        + [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p],
    ),
    "nqa_tp_scatter_bwd_x_paired": (
        c_int32,
        [c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]
        + [c_int64, c_int64, c_void_p, c_int64, c_void_p],
    ),
    "nqa_edge_vectors_fwd": (
        c_int32,
        [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p],
    ),
    "nqa_edge_vectors_bwd": (
        c_int32,
        [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_double, c_void_p, c_void_p, c_void_p],
    ),
    "nqa_virial_finalize": (c_int32, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p]),
    "nqa_edge_embed_fwd": (
        c_int32,
        [c_int32, c_int32, c_void_p, c_int64, c_double, c_void_p, c_int32, c_void_p, c_double, c_double]
        + [c_void_p, c_void_p, c_void_p, c_void_p],
    ),
    "nqa_edge_embed_bwd": (
        c_int32,
        [c_int32, c_int32, c_void_p, c_int64, c_double, c_void_p, c_int32, c_void_p, c_double, c_double]
        + [c_void_p, c_void_p, c_void_p, c_void_p],
    ),
    "nqa_edge_embed_fwd_paired": (
        c_int32,
        [c_int32, c_int32, c_void_p, c_int64, c_double, c_int32, c_void_p, c_double, c_double, c_void_p, c_int64]
        + [c_void_p, c_void_p, c_void_p, c_void_p],
    ),
    "nqa_edge_embed_bwd_paired": (
        c_int32,
        [c_int32, c_int32, c_void_p, c_int64, c_double, c_int32, c_void_p, c_double, c_double, c_void_p, c_int64]
        + [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    ),
    "nqa_edge_embed_bwd_bwd": (
        c_int32,
        [c_int32, c_int32, c_void_p, c_int64, c_double, c_void_p, c_int32, c_void_p, c_double, c_double]
        + [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    ),
    "nqa_radial_mlp_supported": (c_int32, [c_int32, c_int32, c_int32, c_int32]),
    "nqa_radial_mlp_workspace_bytes": (c_int64, [c_int32, c_int32, c_int32, c_int32]),
    "nqa_radial_mlp_fwd": (
        c_int32,
        [c_int32, c_int32, c_void_p, c_void_p, c_double, c_void_p, c_double, c_int32, c_int32, c_int32, c_int64]
        + [c_void_p, c_void_p, c_int64, c_int32, c_void_p],
    ),
    "nqa_radial_mlp_bwd": (
        c_int32,
        [c_int32, c_int32, c_void_p, c_void_p, c_double, c_void_p, c_double, c_void_p, c_int32, c_int32, c_int32]
        + [c_int64, c_void_p, c_void_p, c_int64, c_int32, c_void_p],
    ),
    "nqa_node_linear": (
        c_int32,
        [c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_int32, c_int32]
        + [c_int64, c_int32, c_int32, c_int64, c_double, c_int32, c_void_p],
    ),
    "nqa_node_weights_pack_bytes": (c_int64, [c_void_p, c_int32, c_void_p, c_int32, c_int32]),
    "nqa_node_weights_pack": (
        c_int32, [c_void_p, c_void_p, c_int32, c_void_p, c_int32, c_int32, c_int64, c_void_p, c_void_p],
    ),
    "nqa_node_linear_packed": (
        c_int32,
        [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_int32, c_int32, c_int32, c_int32]
        + [c_int64, c_double, c_void_p],
    ),
    "nqa_node_linear_ordered": (
        c_int32,
        [c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_int32, c_int32,
         c_int64, c_int32, c_int32, c_int64, c_double, c_int32, c_void_p],
    ),
    "nqa_node_linear_packed_ordered": (
        c_int32,
        [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_int32, c_int32, c_int32]
        + [c_int32, c_int64, c_double, c_void_p],
    ),
    "nqa_node_fused": (
        c_int32, [c_void_p, c_int32, c_void_p, c_void_p, c_int64, c_void_p, c_int32, c_void_p, c_int32, c_void_p],
    ),
    "nqa_node_fused_plan": (c_int32, [c_void_p, c_int32, c_void_p, c_int32, _P32, c_int32, _P32, c_int32]),
    "nqa_radial_mlp_last_fwd": (
        c_int32,
        [c_int32, c_int32, c_void_p, c_void_p, c_double, c_int32, c_int32, c_int64, c_void_p, c_void_p, c_int64, c_int32,
         c_void_p],
    ),
    "nqa_radial_mlp_last_bwd": (
        c_int32,
        [c_int32, c_int32, c_void_p, c_void_p, c_double, c_void_p, c_void_p, c_int32, c_int32, c_int64, c_void_p, c_void_p,
         c_int64, c_int32, c_void_p],
    ),
    "nqa_energy_head": (
        c_int32,
        [c_int32, c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_int32, c_int32,
         c_double, c_int64, c_void_p],
    ),
    "nqa_neighbor_list_workspace_bytes": (c_int64, [c_int64]),
    "nqa_neighbor_list_count": (
        c_int32,
        [c_void_p, c_void_p, c_void_p, c_double, c_int64, c_void_p, c_int64, c_void_p, c_void_p],
    ),
    "nqa_neighbor_list_fill": (c_int32, [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p]),
    "nqa_neighbor_list_fill_padded": (
        c_int32, [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "nqa_gate": (
        c_int32,
        [c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int64, c_void_p],
    ),
    "nqa_radial_mlp_bwd_paired": (
        c_int32,
        [c_int32, c_int32, c_void_p, c_void_p, c_double, c_void_p, c_double, c_void_p, c_void_p, c_int32, c_int32,
         c_int32, c_int64, c_void_p, c_void_p, c_int64, c_int32, c_void_p],
    ),
    "nqa_radial_mlp_train_tiles": (c_int64, [c_int64]),
    "nqa_radial_mlp_bwd_train": (
        c_int32,
        [c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_double, c_void_p, c_double, c_void_p, c_int32, c_int32,
         c_int32, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_void_p],
    ),
    "nqa_radial_mlp_fwd_tangent": (
        c_int32,
        [c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_double, c_void_p, c_double, c_int32, c_int32, c_int32,
         c_int64, c_void_p, c_void_p, c_int64, c_int32, c_void_p],
    ),
    "nqa_wgrad_splits": (c_int32, [c_void_p, c_int32, c_int32, c_int64]),
    "nqa_wgrad": (
        c_int32,
        [c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int64, c_int64, c_int64, c_int32, c_int64, c_int32,
         c_void_p, c_void_p],
    ),
}

_lock = threading.Lock()
_lib = None


class NequipAmdLibraryError(RuntimeError):
    pass


def load() -> ctypes.CDLL:
    """Load (once) and return the shared library; raise loudly if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise NequipAmdLibraryError(
                f"{LIB_PATH} not found: the HIP kernels have not been built. Run "
                "`python -m nequip_amd.csrc.build` (or `__graft_entry__.build()`); there is no CPU fallback."
            )
        lib = ctypes.CDLL(LIB_PATH)
        for name, (restype, argtypes) in SIGNATURES.items():
            try:
                fn = getattr(lib, name)
            except AttributeError as exc:  # pragma: no cover
                raise NequipAmdLibraryError(f"{LIB_PATH} does not export `{name}`") from exc
            fn.restype = restype
            fn.argtypes = argtypes
        if lib.nqa_abi_version() != 1:
            raise NequipAmdLibraryError("libnequip_amd.so ABI version mismatch; rebuild the library")
        _lib = lib
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != NQA_OK:
        msg = load().nqa_last_error()
        raise RuntimeError(f"libnequip_amd {what} failed (code {rc}): {msg.decode() if msg else ''}")


def int32_array(values):
    arr = (c_int32 * max(len(values), 1))()
    for i, v in enumerate(values):
        arr[i] = int(v)
    return arr
