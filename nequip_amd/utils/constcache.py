"""Derived constants of weight tensors (packed / transposed / split images), cached on the identity of the weight's STORAGE.

The eager modules keep their weights as persistent tensors and the derived images used to ride on those Python objects.
A compiled graph hands the same constants to the dispatcher ops as fresh tensor objects at every call (the AOTInductor
runtime wraps its constant buffers anew), so the cache is keyed on what stays the same: the storage (held by the entry, so
its address cannot be recycled under it), data pointer, version counter, shape, strides and dtype.  A tensor that is
recomputed at every call (a graph whose weight preparation was not folded, ``utils/aot.py::fold_constants``) never hits --
its storage is new each time -- and only costs an entry of the bounded LRU.  As everywhere in this package, a weight that
is rewritten in place behind PyTorch's back (``p.data.copy_()``) needs ``clear()`` /
``module.invalidate_weight_cache()``."""

from __future__ import annotations

from collections import OrderedDict
from typing import Callable

import torch

MAX_ENTRIES = 512
_entries: "OrderedDict[tuple, tuple]" = OrderedDict()


def _version(t: torch.Tensor) -> int:
    try:
        return t._version
    except RuntimeError:  # inference tensors do not track a version counter
        return -1


def get(t: torch.Tensor, tag, build: Callable[[], object]):
    """``build()`` once per (tensor identity as described above, tag)."""
    st = t.untyped_storage()
    key = (st._cdata, t.data_ptr(), _version(t), tuple(t.shape), tuple(t.stride()), t.dtype, tag)
    hit = _entries.get(key)
    if hit is not None:
        _entries.move_to_end(key)
        return hit[1]
    val = build()
    _entries[key] = (st, val)
    while len(_entries) > MAX_ENTRIES:
        _entries.popitem(last=False)
    return val


def clear() -> None:
    _entries.clear()
