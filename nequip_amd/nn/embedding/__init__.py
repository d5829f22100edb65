from ._edge import EdgeLengthNormalizer, BesselEdgeLengthEncoding, SphericalHarmonicEdgeAttrs  # noqa: F401
from .cutoffs import PolynomialCutoff  # noqa: F401
from .node import NodeTypeEmbed  # noqa: F401
