"""Configuration dataclasses of the drop-in API.

Field names and defaults ARE the public API of the reference (``CodebookParams`` is what callers pass to
``VectorQuantize(codebook_params=...)``), so they are kept identical:
/root/reference/vector_quantization/codebooks.py:31-78.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Optional


@dataclass
class AffineParameters:
    """Running-statistics affine re-parameterisation (training-time; not on the search path)."""

    sync: bool
    batch_decay: float = 0.99
    codebook_decay: float = 0.9


@dataclass
class KmeansParameters:
    """Lloyd iterations used to seed the codebook from the first batch."""

    iter: int = 10
    sync: bool = True


@dataclass
class GumbelParams:
    """Sampling options of the code selection: deterministic argmax (native, bit-exact) or stochastic Gumbel-max sampling
    (native similarities + device RNG); the straight-through / reinmax relaxations are not provided."""

    temperature: float = 1.0
    stochastic: bool = False
    reinmax: bool = False
    straight_through: bool = False
    dim: int = -1
    training: bool = True


@dataclass
class CodebookParams:
    dim: int
    codebook_size: int
    num_codebooks: int = 1
    initialization_by_kmeans: bool = False
    kmeans_params: Optional[KmeansParameters] = None
    decay: float = 0.8
    eps_for_smoothing: float = 1e-5
    threshold_ema_dead_code: int = 2
    reset_cluster_size: Optional[int] = None
    use_ddp: bool = False
    distributed_replace_codes: bool = True
    learnable_codebook: bool = False
    gumbel_params: GumbelParams = field(default_factory=GumbelParams)
    ema_update: bool = True
    use_affine: bool = False
    affine_params: Optional[AffineParameters] = None
    transform_input: str = "identity"
    use_cosine_sim: bool = False
    weights_regularization: str = "identity"
