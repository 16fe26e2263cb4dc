"""MI355X-native drop-in for the nearest-codebook hot path of MisterBourbaki/vector-quantization-by-ml.

Import surface kept from the reference (``vector_quantization/__init__.py:11-12,16-28``):

    from vector_quantization import VectorQuantize, ResidualVQ, GroupedResidualVQ
    from vector_quantization.codebooks import CodebookParams, KmeansParameters, GumbelParams, AffineParameters, Codebook

The search itself (distance -> first argmax -> gather, straight-through, squared error, residual loop) is
hand-written HIP for gfx950 behind the C ABI in ``include/vq_mi355x.h``; there is no PyTorch/CPU fallback.
The reference's other quantizer families (FSQ, LFQ, latent quantization and their residual variants) never
touch the codebook search and are not part of this build.
"""
from . import ops  # noqa: F401  (registers torch.ops.vq_mi355x.*)
from .codebook import Codebook
from .graphs import GraphedForward
from .params import AffineParameters, CodebookParams, GumbelParams, KmeansParameters
from .projection import RandomProjectionQuantizer
from .quantizer import LossBreakdown, VectorQuantize
from .residual import GroupedResidualVQ, ResidualVQ
from .sharded import ShardedCodebookSearch

__all__ = [
    "AffineParameters",
    "Codebook",
    "CodebookParams",
    "GraphedForward",
    "GroupedResidualVQ",
    "GumbelParams",
    "KmeansParameters",
    "LossBreakdown",
    "RandomProjectionQuantizer",
    "ResidualVQ",
    "ShardedCodebookSearch",
    "VectorQuantize",
]
