"""Import-path shim: ``from vector_quantization.residual_vq import ResidualVQ, GroupedResidualVQ``."""
from .residual import GroupedResidualVQ, ResidualVQ  # noqa: F401
