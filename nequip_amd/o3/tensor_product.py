"""Descriptor of an e3nn-style ``'uvu'`` tensor product and its native plan.

The reference builds ``e3nn.o3.TensorProduct(irreps_in1, irreps_in2, irreps_out, instructions,
shared_weights=False, internal_weights=False)`` inside ``TensorProductScatter.__init__``
(``nequip/nn/_tp_scatter_base.py:24-31``) and reads ``self.tp.weight_numel`` from it
(``nequip/nn/interaction_block.py:121``).  This module keeps that object *shape*: a parameter-free
``torch.nn.Module`` named ``tp`` carrying the irreps, the instruction list and ``weight_numel`` --
but it owns no arithmetic.  The arithmetic lives in the HIP kernels, reached through the native plan
(``nqa_plan_create``, ``include/nequip_amd.h``) that is created here from exactly the constructor
arguments.
"""

from __future__ import annotations

import ctypes
from typing import List, NamedTuple, Optional, Sequence, Tuple

import torch

from .. import _lib
from .irreps import Irreps


class Instruction(NamedTuple):
    i_in1: int
    i_in2: int
    i_out: int
    connection_mode: str
    has_weight: bool
    path_weight: float = 1.0


def _normalize_instructions(instructions: Sequence) -> List[Instruction]:
    out = []
    for ins in instructions:
        ins = tuple(ins)
        if len(ins) == 5:
            ins = ins + (1.0,)
        i1, i2, io, mode, has_weight, pw = ins
        out.append(Instruction(int(i1), int(i2), int(io), str(mode), bool(has_weight), float(pw)))
    return out


class NativePlan:
    """RAII wrapper of ``nqa_plan`` plus its device-table image (host bytes)."""

    def __init__(
        self,
        irreps_in1: Irreps,
        irreps_in2: Irreps,
        irreps_out: Irreps,
        instructions: Sequence[Instruction],
        layout_in1: int = _lib.NQA_LAYOUT_MUL_IR,
        layout_out: int = _lib.NQA_LAYOUT_MUL_IR,
    ):
        lib = _lib.load()
        self._lib = lib
        self._handle = ctypes.c_void_p()
        # what the native object was built from: copies / pickles rebuild it (a raw handle cannot be shared or saved)
        self._ctor_args = (irreps_in1, irreps_in2, irreps_out, list(instructions), layout_in1, layout_out)

        def arrs(irreps):
            return (
                _lib.int32_array([mul for mul, _ in irreps]),
                _lib.int32_array([ir.l for _, ir in irreps]),
                _lib.int32_array([ir.p for _, ir in irreps]),
            )

        a1, a2, ao = arrs(irreps_in1), arrs(irreps_in2), arrs(irreps_out)
        n = len(instructions)
        pw = (ctypes.c_double * max(n, 1))(*[ins.path_weight for ins in instructions])
        rc = lib.nqa_plan_create(
            len(irreps_in1), *a1,
            len(irreps_in2), *a2,
            len(irreps_out), *ao,
            n,
            _lib.int32_array([ins.i_in1 for ins in instructions]),
            _lib.int32_array([ins.i_in2 for ins in instructions]),
            _lib.int32_array([ins.i_out for ins in instructions]),
            pw,
            layout_in1,
            layout_out,
            ctypes.byref(self._handle),
        )  # fmt: skip
        _lib.check(rc, "nqa_plan_create")
        nbytes = lib.nqa_plan_image_bytes(self._handle)
        buf = (ctypes.c_uint8 * nbytes)()
        _lib.check(lib.nqa_plan_image_write(self._handle, buf, nbytes), "nqa_plan_image_write")
        self.image = torch.frombuffer(bytearray(buf), dtype=torch.uint8).clone()

    @property
    def handle(self) -> ctypes.c_void_p:
        return self._handle

    def __reduce__(self):
        # copy.deepcopy(model), torch.save(model): a fresh native plan from the same description
        return (NativePlan, self._ctor_args)

    def query(self, field: int) -> int:
        return int(self._lib.nqa_plan_query(self._handle, field))

    def __del__(self):
        try:
            if self._handle:
                self._lib.nqa_plan_destroy(self._handle)
                self._handle = ctypes.c_void_p()
        except Exception:  # pragma: no cover  (interpreter shutdown)
            pass


class TensorProduct(torch.nn.Module):
    """Parameter-free descriptor with the attribute surface of ``e3nn.o3.TensorProduct`` that nequip reads."""

    def __init__(
        self,
        irreps_in1,
        irreps_in2,
        irreps_out,
        instructions: Sequence,
        shared_weights: bool = False,
        internal_weights: bool = False,
    ):
        super().__init__()
        if shared_weights or internal_weights:
            raise NotImplementedError(
                "only the per-edge-weight form used by nequip (shared_weights=False, internal_weights=False)"
            )
        self.irreps_in1 = Irreps(irreps_in1)
        self.irreps_in2 = Irreps(irreps_in2)
        self.irreps_out = Irreps(irreps_out)
        self.instructions = _normalize_instructions(instructions)
        for ins in self.instructions:
            if ins.connection_mode != "uvu":
                raise NotImplementedError(f"connection mode {ins.connection_mode!r}; nequip uses 'uvu'")
            if not ins.has_weight:
                raise NotImplementedError("'uvu' instructions without weights are not used by nequip")
            mul1 = self.irreps_in1[ins.i_in1].mul
            if mul1 != self.irreps_out[ins.i_out].mul:
                raise ValueError("'uvu' needs mul_in1 == mul_out")
        self.shared_weights = False
        self.internal_weights = False
        self.weight_numel = sum(
            self.irreps_in1[ins.i_in1].mul * self.irreps_in2[ins.i_in2].mul for ins in self.instructions
        )

    def weight_slices(self) -> List[Tuple[int, int]]:
        out, off = [], 0
        for ins in self.instructions:
            n = self.irreps_in1[ins.i_in1].mul * self.irreps_in2[ins.i_in2].mul
            out.append((off, off + n))
            off += n
        return out

    def forward(self, *args, **kwargs):  # pragma: no cover
        raise RuntimeError(
            "nequip_amd.o3.TensorProduct is a descriptor; the product is evaluated by the fused HIP "
            "TensorProductScatter kernels (there is no unfused / CPU path)"
        )

    def extra_repr(self) -> str:
        return (
            f"{self.irreps_in1} x {self.irreps_in2} -> {self.irreps_out} | "
            f"{len(self.instructions)} paths | {self.weight_numel} weights"
        )
