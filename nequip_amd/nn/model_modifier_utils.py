"""Model-modifier plumbing: the reference's plugin seam (``nequip/nn/model_modifier_utils.py:22-107``),
restated so that accelerated modules can be swapped in the same way with or without nequip installed."""

from typing import Callable, Final, List, Optional

import torch

_MODEL_MODIFIER_PERSISTENT_ATTR_NAME: Final[str] = "_nequip_model_modifier_is_persistent"
_MODEL_MODIFIER_PRIVATE_ATTR_NAME: Final[str] = "_nequip_model_modifier_is_private"
_MODEL_MODIFIER_UNSUPPORTED_DEVICES_ATTR_NAME: Final[str] = "_nequip_model_modifier_unsupported_devices"
_MODEL_MODIFIER_SUPPORTED_COMPILE_MODES_ATTR_NAME: Final[str] = "_nequip_model_modifier_supported_compile_modes"


def model_modifier(persistent: bool, private: Optional[bool] = None, unsupported_devices: List[str] = [],
                   supported_compile_modes: Optional[List[str]] = None):
    def decorator(func):
        assert isinstance(func, classmethod), "@model_modifier must be applied after @classmethod"
        setattr(func.__func__, _MODEL_MODIFIER_PERSISTENT_ATTR_NAME, persistent)
        if private is not None:
            setattr(func.__func__, _MODEL_MODIFIER_PRIVATE_ATTR_NAME, private)
        setattr(func.__func__, _MODEL_MODIFIER_UNSUPPORTED_DEVICES_ATTR_NAME, unsupported_devices)
        setattr(func.__func__, _MODEL_MODIFIER_SUPPORTED_COMPILE_MODES_ATTR_NAME, supported_compile_modes)
        return func

    return decorator


def is_model_modifier(func: Callable) -> bool:
    return hasattr(func, _MODEL_MODIFIER_PERSISTENT_ATTR_NAME)


def replace_submodules(model: torch.nn.Module, target_cls: type,
                       factory: Callable[[torch.nn.Module], torch.nn.Module]) -> torch.nn.Module:
    for name, child in list(model.named_children()):
        if isinstance(child, target_cls):
            model._modules[name] = factory(child)
        else:
            replace_submodules(child, target_cls, factory)
    return model
