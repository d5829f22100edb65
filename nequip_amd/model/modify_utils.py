"""Applying model modifiers by name (mirror of ``nequip.model.modify`` / ``get_all_modifiers``,
``nequip/model/modify_utils.py:35-137``, for already built ``torch.nn.Module`` models): modifiers are
``@model_modifier``-decorated classmethods of modules present in the model, discovered by walking the module tree,
addressed by their globally unique name and applied in the given order."""

from __future__ import annotations

import inspect
from typing import Any, Callable, Dict, List, Optional

import torch

from ..nn.model_modifier_utils import is_model_modifier


def get_all_modifiers(module: torch.nn.Module, _all: Optional[Dict[str, Callable]] = None) -> Dict[str, Callable]:
    if _all is None:
        _all = {}
    for name, member in inspect.getmembers(module, predicate=inspect.ismethod):
        if is_model_modifier(member):
            if name in _all:
                assert _all[name] == member, (
                    f"Found at least two non-unique modifiers with same name `{name}`: {_all[name]!r} and {member!r}"
                )
            _all[name] = member
    for _, child in module.named_children():
        get_all_modifiers(child, _all)
    return _all


def modify(model: torch.nn.Module, modifiers: List[Dict[str, Any]]) -> torch.nn.Module:
    """``modifiers``: list of ``{"modifier": name, **kwargs}``; unknown names raise ``RuntimeError`` listing the
    registered ones (same behaviour and message as the reference)."""
    if not isinstance(model, torch.nn.Module):
        raise RuntimeError("Unrecognized model object found.")
    assert isinstance(modifiers, list)
    avail = get_all_modifiers(model)
    for cfg in modifiers:
        cfg = dict(cfg)
        name = cfg.pop("modifier")
        if name not in avail:
            raise RuntimeError(
                f"`{name}` is not a registered model modifier. The following are registered model modifiers: {list(avail)}"
            )
        model = avail[name](model, **cfg)
    return model
