"""``RandomProjectionQuantizer`` (BEST-RQ, https://arxiv.org/abs/2202.01855): a fixed random projection followed by a
multi-head cosine-similarity codebook search in eval mode -- a pure CONSUMER of the native search.

The fork's constructor is broken as shipped (it passes ``codebook_size=`` / ``use_cosine_sim=`` keyword arguments that
``VectorQuantize.__init__`` no longer accepts, /root/reference/vector_quantization/random_projection_quantizer.py:30-37);
this is the repaired equivalent with the same signature: the cosine codebook is expressed through ``CodebookParams``
(l2-normalised inputs and codes, so the dot product IS the cosine similarity the paper uses).
"""
from __future__ import annotations

import torch
from torch import nn

from .params import CodebookParams
from .quantizer import VectorQuantize


class RandomProjectionQuantizer(nn.Module):
    def __init__(self, *, dim, codebook_size, codebook_dim, num_codebooks=1, norm=True, **kwargs):
        super().__init__()
        self.num_codebooks = num_codebooks
        rand_projs = torch.empty(num_codebooks, dim, codebook_dim)
        nn.init.xavier_normal_(rand_projs)
        self.register_buffer("rand_projs", rand_projs)
        # section 3 of the paper: inputs are normalised to zero mean / unit variance to prevent collapse
        self.norm = nn.LayerNorm(dim, elementwise_affine=False) if norm else nn.Identity()
        params = CodebookParams(dim=codebook_dim, codebook_size=codebook_size, use_cosine_sim=True,
                                transform_input="l2norm", weights_regularization="l2norm")
        self.vq = VectorQuantize(dim=codebook_dim * num_codebooks, codebook_params=params, codebook_dim=codebook_dim,
                                 heads=num_codebooks, separate_codebook_per_head=True, **kwargs)

    def forward(self, x, indices=None):
        """-> code indices [b, n, num_codebooks] (or [b, n]); with ``indices`` the cross entropy of the similarities
        against them (random_projection_quantizer.py:40-60)."""
        x = self.norm(x)
        x = torch.einsum("bnd,hde->bnhe", x, self.rand_projs)
        x = x.reshape(x.shape[0], x.shape[1], -1)
        self.vq.eval()
        if indices is not None:
            _, ce_loss = self.vq(x, indices=indices)
            return ce_loss
        _, found, _ = self.vq(x)
        return found
