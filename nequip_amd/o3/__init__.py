from .irreps import Irrep, Irreps  # noqa: F401
from .wigner import wigner_3j  # noqa: F401
