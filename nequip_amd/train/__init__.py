from .simple_ddp import SimpleDDPStrategy, all_reduce_gradients, broadcast_parameters  # noqa: F401
