from ._edge import (EdgeLengthNormalizer, BesselEdgeLengthEncoding, SphericalHarmonicEdgeAttrs,  # noqa: F401
                    cutoff_partialdict_to_tensor)
from .cutoffs import PolynomialCutoff  # noqa: F401
from .node import NodeTypeEmbed  # noqa: F401
