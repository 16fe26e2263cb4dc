"""Import-path shim: ``from vector_quantization.random_projection_quantizer import RandomProjectionQuantizer``."""
from .projection import RandomProjectionQuantizer  # noqa: F401
