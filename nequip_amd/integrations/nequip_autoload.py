"""Target of the ``nequip.extension`` / ``init_always`` entry point (``pyproject.toml``).

nequip loads this module while ``import nequip`` is still running (``nequip/__init__.py:23-38``).  At that moment
``nequip.nn`` may or may not be importable without a cycle, so registration is attempted right away and, if nequip is not
ready yet, retried lazily: the first ``import`` of a target module by anybody triggers it through a meta-path finder that
removes itself once every target is served.  Two modifiers are attached:

* ``enable_NequipAMD`` on ``nequip.nn._tp_scatter_base.TensorProductScatter`` (``nequip_extension.py``: the tensor product /
  scatter alone, the seam OpenEquivariance and cuEquivariance use);
* ``enable_NequipAMD_full`` on ``nequip.nn.convnetlayer.ConvNetLayer`` (``nequip_full.py``: every module of the benchmarked
  path, so that the fused edge embedding, paired radial MLP, node stage and energy head run on a model nequip built).

Explicit ``nequip_extension.register()`` / ``nequip_full.register_full()`` always work as well."""

from __future__ import annotations

import importlib.abc
import importlib.util
import sys

_TARGET = "nequip.nn._tp_scatter_base"


def _register_tps(cls) -> None:
    from .nequip_extension import register

    register(cls)


def _register_full(cls) -> None:
    from .nequip_full import register_full

    register_full(cls)


# module -> (class that carries the modifier, how to attach it)
_TARGETS = {
    _TARGET: ("TensorProductScatter", _register_tps),
    "nequip.nn.convnetlayer": ("ConvNetLayer", _register_full),
}
_done = set()


def _try_register() -> bool:
    for name, (cls_name, reg) in _TARGETS.items():
        if name in _done:
            continue
        mod = sys.modules.get(name)
        cls = getattr(mod, cls_name, None) if mod is not None else None
        if cls is not None:
            reg(cls)
            _done.add(name)
    return len(_done) == len(_TARGETS)


class _RegisterAfterImport(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    """Lets the normal machinery import a target module and attaches its modifier right after."""

    def __init__(self):
        self._busy = False
        self._inner = {}

    def find_spec(self, fullname, path, target=None):
        if fullname not in _TARGETS or fullname in _done or self._busy:
            return None
        self._busy = True
        try:
            spec = importlib.util.find_spec(fullname)
        finally:
            self._busy = False
        if spec is None or spec.loader is None:
            return None
        self._inner[fullname] = spec.loader
        spec.loader = self
        return spec

    def create_module(self, spec):
        inner = self._inner[spec.name]
        return inner.create_module(spec) if hasattr(inner, "create_module") else None

    def exec_module(self, module):
        name = module.__name__
        self._inner[name].exec_module(module)
        cls_name, reg = _TARGETS[name]
        cls = getattr(module, cls_name, None)
        if cls is not None:
            reg(cls)
            _done.add(name)
        if len(_done) == len(_TARGETS):  # every target served: the finder leaves
            try:
                sys.meta_path.remove(self)
            except ValueError:
                pass


def install() -> None:
    if _try_register():
        return
    if not any(isinstance(f, _RegisterAfterImport) for f in sys.meta_path):
        sys.meta_path.insert(0, _RegisterAfterImport())


install()
