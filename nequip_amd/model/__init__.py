from .nequip_models import FullNequIPGNNModel, NequIPGNNModel, PresetNequIPGNNModel  # noqa: F401
