"""Minimal, dependency-free ``Irrep`` / ``Irreps`` for the NequIP hot path.

Host-side mirror of the subset of ``e3nn.o3.Irreps`` that the reference's hot path touches
(reference call sites: ``nequip/nn/interaction_block.py:89-109`` builds the ``uvu`` instruction list from
``ir_in * ir_edge``, ``.sort()`` and ``.simplify()``; ``nequip/nn/convnetlayer.py:71-117`` filters
irreps with ``ir in irreps``; ``nequip/model/nequip_models.py:161-190`` uses
``Irreps.spherical_harmonics`` and ``repr``; ``tests/unit/nn/test_tp_scatter_kernel.py:141-142`` uses
``irreps.randn(n, -1)``).

Semantics restated from e3nn 0.6.x: an ``Irrep`` is the pair ``(l, p)`` with
``p in {+1 'e', -1 'o'}``; ordering is plain tuple order on ``(l, p)`` (so ``0o < 0e < 1o < 1e``);
the flattened feature layout is ``mul_ir``: blocks concatenated in list order, each block shaped
``[mul, 2l+1]`` with the ``m`` index fastest.
"""

from __future__ import annotations

import re
from typing import Iterator, List, NamedTuple, Sequence, Tuple, Union

import torch


class Irrep(tuple):
    """Irreducible representation of O(3): ``(l, p)``."""

    def __new__(cls, l, p=None):
        if p is None:
            if isinstance(l, Irrep):
                return l
            if isinstance(l, str):
                name = l.strip()
                m = re.fullmatch(r"(\d+)([eoy])", name)
                if m is None:
                    raise ValueError(f"unable to convert string '{name}' into an Irrep")
                ll = int(m.group(1))
                p = {"e": 1, "o": -1, "y": (-1) ** ll}[m.group(2)]
                l = ll
            elif isinstance(l, tuple):
                l, p = l
            else:
                raise ValueError(f"unable to convert {l!r} into an Irrep")
        if not isinstance(l, int) or l < 0:
            raise ValueError(f"l must be a non-negative integer, got {l!r}")
        if p not in (-1, 1):
            raise ValueError(f"parity must be +1 or -1, got {p!r}")
        return super().__new__(cls, (l, p))

    @property
    def l(self) -> int:  # noqa: E743
        return self[0]

    @property
    def p(self) -> int:
        return self[1]

    @property
    def dim(self) -> int:
        return 2 * self.l + 1

    def __repr__(self) -> str:
        return f"{self.l}{'e' if self.p == 1 else 'o'}"

    def __mul__(self, other) -> Iterator["Irrep"]:
        """Selection rule: ``|l1-l2| <= l <= l1+l2`` in increasing ``l``, parity ``p1*p2``."""
        other = Irrep(other)
        p = self.p * other.p
        for l in range(abs(self.l - other.l), self.l + other.l + 1):
            yield Irrep(l, p)

    def __rmul__(self, mul: int) -> "Irreps":
        assert isinstance(mul, int)
        return Irreps([(mul, self)])


class _MulIr(NamedTuple):
    mul: int
    ir: Irrep

    @property
    def dim(self) -> int:
        return self.mul * self.ir.dim

    def __repr__(self) -> str:
        return f"{self.mul}x{self.ir}"


class _SortResult(NamedTuple):
    irreps: "Irreps"
    p: Tuple[int, ...]
    inv: Tuple[int, ...]


class Irreps(tuple):
    """Direct sum of irreps with multiplicities, e.g. ``Irreps("64x0e + 64x1o")``."""

    def __new__(cls, irreps=None):
        if isinstance(irreps, Irreps):
            return super().__new__(cls, irreps)
        out: List[_MulIr] = []
        if irreps is None:
            pass
        elif isinstance(irreps, Irrep):
            out.append(_MulIr(1, irreps))
        elif isinstance(irreps, str):
            s = irreps.strip()
            if s != "":
                for term in s.split("+"):
                    term = term.strip()
                    if "x" in term:
                        mul_s, ir_s = term.split("x")
                        out.append(_MulIr(int(mul_s), Irrep(ir_s)))
                    else:
                        out.append(_MulIr(1, Irrep(term)))
        else:
            for item in irreps:
                if isinstance(item, _MulIr):
                    mul, ir = item
                elif isinstance(item, Irrep):
                    mul, ir = 1, item
                elif isinstance(item, str):
                    mul, ir = 1, Irrep(item)
                elif len(item) == 2:
                    mul, ir = item
                    ir = Irrep(ir)
                else:
                    raise ValueError(f"unable to interpret {item!r} as (mul, irrep)")
                if not isinstance(mul, int) or mul < 0:
                    raise ValueError(f"multiplicity must be a non-negative int, got {mul!r}")
                out.append(_MulIr(mul, ir))
        return super().__new__(cls, out)

    # --- constructors ---------------------------------------------------------------------------
    @staticmethod
    def spherical_harmonics(lmax: int, p: int = -1) -> "Irreps":
        return Irreps([(1, (l, p**l)) for l in range(lmax + 1)])

    # --- basic properties -----------------------------------------------------------------------
    @property
    def dim(self) -> int:
        return sum(mul * ir.dim for mul, ir in self)

    @property
    def num_irreps(self) -> int:
        return sum(mul for mul, _ in self)

    @property
    def ls(self) -> List[int]:
        return [ir.l for mul, ir in self for _ in range(mul)]

    @property
    def lmax(self) -> int:
        if len(self) == 0:
            raise ValueError("cannot get lmax of empty Irreps")
        return max(ir.l for _, ir in self)

    def slices(self) -> List[slice]:
        s, i = [], 0
        for mul_ir in self:
            s.append(slice(i, i + mul_ir.dim))
            i += mul_ir.dim
        return s

    def offsets(self) -> List[int]:
        return [s.start for s in self.slices()]

    # --- algebra --------------------------------------------------------------------------------
    def simplify(self) -> "Irreps":
        """Merge *adjacent* equal irreps and drop zero multiplicities (no sorting)."""
        out: List[Tuple[int, Irrep]] = []
        for mul, ir in self:
            if mul == 0:
                continue
            if out and out[-1][1] == ir:
                out[-1] = (out[-1][0] + mul, ir)
            else:
                out.append((mul, ir))
        return Irreps(out)

    def sort(self) -> _SortResult:
        """Stable sort by ``(l, p)`` tuple order; ``p[i_old] = i_new``, ``inv[i_new] = i_old``."""
        out = sorted((ir, i, mul) for i, (mul, ir) in enumerate(self))
        inv = tuple(i for _, i, _ in out)
        p = [0] * len(inv)
        for new, old in enumerate(inv):
            p[old] = new
        return _SortResult(Irreps([(mul, ir) for ir, _, mul in out]), tuple(p), inv)

    def remove_zero_multiplicities(self) -> "Irreps":
        return Irreps([(mul, ir) for mul, ir in self if mul > 0])

    def count(self, ir) -> int:
        ir = Irrep(ir)
        return sum(mul for mul, ir2 in self if ir2 == ir)

    def __contains__(self, ir) -> bool:
        try:
            ir = Irrep(ir)
        except (ValueError, TypeError):
            return False
        return any(ir == ir2 for _, ir2 in self)

    def __add__(self, other) -> "Irreps":
        return Irreps(tuple(self) + tuple(Irreps(other)))

    def __mul__(self, n: int) -> "Irreps":
        if not isinstance(n, int):
            return NotImplemented
        return Irreps(tuple(self) * n)

    __rmul__ = __mul__

    def __getitem__(self, i):
        x = super().__getitem__(i)
        if isinstance(i, slice):
            return Irreps(x)
        return x

    def __repr__(self) -> str:
        return "+".join(f"{mul_ir}" for mul_ir in self)

    # --- data -----------------------------------------------------------------------------------
    def randn(self, *size: int, dtype=None, device=None, requires_grad: bool = False, generator=None):
        """Standard-normal features; exactly one entry of ``size`` must be ``-1`` (the irreps axis)."""
        di = size.index(-1)
        shape = size[:di] + (self.dim,) + size[di + 1 :]
        if dtype is None:
            dtype = torch.get_default_dtype()
        return torch.randn(*shape, dtype=dtype, device=device, generator=generator).requires_grad_(
            requires_grad
        )


IrrepsLike = Union[str, Irreps, Sequence]
