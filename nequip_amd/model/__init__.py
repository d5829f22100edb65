from .nequip_models import FullNequIPGNNModel, NequIPGNNModel, PresetNequIPGNNModel  # noqa: F401
from .modify_utils import get_all_modifiers, modify  # noqa: F401
