"""Import-path shim: ``from vector_quantization.codebooks import CodebookParams, Codebook, ...``."""
from .codebook import Codebook  # noqa: F401
from .params import AffineParameters, CodebookParams, GumbelParams, KmeansParameters  # noqa: F401
