"""Import-path shim: ``from vector_quantization.vector_quantize_pytorch import VectorQuantize``."""
from .quantizer import LossBreakdown, VectorQuantize  # noqa: F401
