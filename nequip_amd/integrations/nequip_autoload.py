"""Target of the ``nequip.extension`` / ``init_always`` entry point (``pyproject.toml``).

nequip loads this module while ``import nequip`` is still running (``nequip/__init__.py:23-38``).  At that moment
``nequip.nn`` may or may not be importable without a cycle, so registration is attempted right away and, if nequip is not
ready yet, retried lazily: the first ``import`` of ``nequip.nn._tp_scatter_base`` by anybody triggers it through a
one-shot meta-path finder.  Explicit ``nequip_amd.integrations.nequip_extension.register()`` always works as well."""

from __future__ import annotations

import importlib.abc
import importlib.util
import sys

_TARGET = "nequip.nn._tp_scatter_base"


def _try_register() -> bool:
    mod = sys.modules.get(_TARGET)
    cls = getattr(mod, "TensorProductScatter", None) if mod is not None else None
    if cls is None:
        return False
    from .nequip_extension import register

    register(cls)
    return True


class _RegisterAfterImport(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    """Lets the normal machinery import ``nequip.nn._tp_scatter_base`` and attaches the modifier right after."""

    def __init__(self):
        self._busy = False

    def find_spec(self, fullname, path, target=None):
        if fullname != _TARGET or self._busy:
            return None
        self._busy = True
        try:
            spec = importlib.util.find_spec(fullname)
        finally:
            self._busy = False
        if spec is None or spec.loader is None:
            return None
        self._inner = spec.loader
        spec.loader = self
        return spec

    def create_module(self, spec):
        return self._inner.create_module(spec) if hasattr(self._inner, "create_module") else None

    def exec_module(self, module):
        self._inner.exec_module(module)
        try:
            sys.meta_path.remove(self)
        except ValueError:
            pass
        cls = getattr(module, "TensorProductScatter", None)
        if cls is not None:
            from .nequip_extension import register

            register(cls)


def install() -> None:
    if _try_register():
        return
    if not any(isinstance(f, _RegisterAfterImport) for f in sys.meta_path):
        sys.meta_path.insert(0, _RegisterAfterImport())


install()
