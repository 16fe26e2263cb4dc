"""PyTorch-CPU restatement of the reference's ATen op sequence.  TEST INFRASTRUCTURE ONLY.

This is the timed ``cpu_baseline`` of bench.py ("port"): the same ATen operators, in the same order, as
the reference's eval-mode ``Codebook.forward`` (citations relative to /root/reference), so that the CPU
number printed next to the MI355X number is what the reference itself would cost on the host cores:

    similarities = -torch.cdist(flatten, embeddings)          codebooks.py:128-129,386
    ind          = similarities.argmax(-1)                     utils/general.py:128
    one_hot      = F.one_hot(ind, K).type(fp32)                utils/general.py:129   (always built)
    quantize     = gather(repeat(embeddings), repeat(ind))     utils/general.py:159-163 (eval)
                 | einsum(one_hot, embeddings)                 codebooks.py:393-395     (train)
    loss         = F.mse_loss(quantize.detach(), x)            vector_quantize_pytorch.py:362 (train)

It is checked against the golden vectors captured from the imported reference
(tests/test_oracle_golden.py) and is never imported by the product package.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


@torch.no_grad()
def codebook_forward(x: torch.Tensor, embeddings: torch.Tensor, use_cosine_sim: bool = False, training: bool = False):
    """x [h, M, D], embeddings [h, K, D] -> (quantize [h,M,D], ind [h,M] int64, similarities [h,M,K])."""
    x = x.float()
    if use_cosine_sim:
        similarities = torch.einsum("h n d, h c d -> h n c", x, embeddings)
    else:
        similarities = -torch.cdist(x, embeddings)
    ind = similarities.argmax(dim=-1)
    one_hot = F.one_hot(ind, embeddings.shape[-2]).type(similarities.dtype)
    if training:
        quantize = torch.einsum("h n c, h c d -> h n d", one_hot, embeddings)
    else:
        d = embeddings.shape[-1]
        quantize = embeddings.gather(1, ind.unsqueeze(-1).expand(-1, -1, d))
    return quantize, ind, similarities


@torch.no_grad()
def vector_quantize_forward(x: torch.Tensor, embeddings: torch.Tensor, use_cosine_sim: bool = False,
                            training: bool = False, commitment_weight: float = 1.0):
    """Single-codebook VectorQuantize on [b, n, D]; returns (quantize, ind [b,n], loss [1])."""
    b, n, d = x.shape
    flat = x.reshape(1, b * n, d)
    q, ind, _ = codebook_forward(flat, embeddings, use_cosine_sim, training)
    q = q.reshape(b, n, d)
    loss = torch.zeros(1)
    if training:
        commit = F.mse_loss(q, x)
        q = x + (q - x)
        loss = loss + commit * commitment_weight
    return q, ind.reshape(b, n), loss


@torch.no_grad()
def residual_vq_forward(x: torch.Tensor, codebooks: torch.Tensor, training: bool = False):
    """codebooks [Q, K, D]; returns (quantized_out [b,n,D], indices [b,n,Q], losses [1,Q])."""
    residual = x
    out = 0.0
    all_ind, all_loss = [], []
    for cb in codebooks:
        q, ind, loss = vector_quantize_forward(residual, cb.unsqueeze(0), training=training)
        residual = residual - q
        out = out + q
        all_ind.append(ind)
        all_loss.append(loss)
    return out, torch.stack(all_ind, dim=-1), torch.stack(all_loss, dim=-1)
