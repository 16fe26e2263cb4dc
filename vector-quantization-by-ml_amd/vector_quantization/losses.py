"""Losses that consume the similarity matrix, built so that ``[M, K]`` never exists for all rows.

The reference hands the full ``similarities [h, M, K]`` tensor (its ``distances``) to three rare consumers:

* cross-entropy of the similarities against code indices -- commitment variant and ``indices=`` teacher forcing
  (vector_quantize_pytorch.py:284-299,338-346);
* the codebook diversity loss: entropy of the batch-averaged softmax (vector_quantize_pytorch.py:324-334);
* the orthogonal regulariser, which only looks at the codebook (utils/losses.py:23-28).

Here the forward of the cross entropy is ONE native sweep with an online-softmax epilogue
(``vq_softmax_stats_f32``: per row log-sum-exp + target logit).  Everything that needs actual matrix entries
(backward passes, the diversity loss) works on bounded row chunks: ``vq_similarities_f32`` emits a chunk, the
chunk's contribution is evaluated with ordinary tensor ops, and gradients flow through ``_SimilarityFn`` whose
backward is ATen's ``_euclidean_dist_backward`` formula (two library GEMMs per chunk).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import search

CHUNK_BYTES = 256 << 20  # bound on one materialised [H, rows, K] fp32 chunk


def _rows_per_chunk(h: int, k: int) -> int:
    rows = CHUNK_BYTES // (4 * max(1, h) * max(1, k))
    return max(128, rows // 128 * 128)


class _SimilarityFn(torch.autograd.Function):
    """sims = -cdist(x, c) (Euclid) or x @ c^T (dot) for x [H, M, D], c [H, K, D].

    Backward (same formulas autograd uses for the reference's ops):
      Euclid  ratio = g / sims (0 where sims == 0);  gx = x * ratio.sum(-1) - ratio @ c;  gc = c * ratio.sum(-2) - ratio^T @ x
              (ATen _euclidean_dist_backward with grad_dist = -g and dist = -sims)
      dot     gx = g @ c;  gc = g^T @ x
    """

    @staticmethod
    def forward(ctx, x, codes, metric, live_codes=None):
        sims = search.get_backend().similarities(x.detach(), codes.detach(), metric=metric)
        ctx.save_for_backward(x, codes, sims)
        ctx.metric = metric
        ctx.live_codes = live_codes  # see `live_codes` in similarity_matrix()
        return sims

    @staticmethod
    def backward(ctx, g):
        x, codes, sims = ctx.saved_tensors
        x, codes = x.detach(), (codes if ctx.live_codes is None else ctx.live_codes).detach()
        gx = gc = None
        if ctx.metric == search.EUCLID:
            ratio = torch.where(sims == 0, torch.zeros_like(g), g / sims)
            if ctx.needs_input_grad[0]:
                gx = x * ratio.sum(-1, keepdim=True) - ratio @ codes
            if ctx.needs_input_grad[1]:
                gc = codes * ratio.sum(-2).unsqueeze(-1) - ratio.transpose(-1, -2) @ x
        else:
            if ctx.needs_input_grad[0]:
                gx = g @ codes
            if ctx.needs_input_grad[1]:
                gc = g.transpose(-1, -2) @ x
        return gx, gc, None, None


def similarity_matrix(x: torch.Tensor, codes: torch.Tensor, metric: int, live_codes=None) -> torch.Tensor:
    """The full [H, M, K] matrix (only for callers that really want the reference's third return value).

    ``live_codes``: in the reference the EMA step overwrites the codebook through ``.data`` right after the similarities
    were computed (codebooks.py:418-425), which autograd does not notice: cdist's backward then multiplies the ratio
    matrix -- computed from the OLD distances -- with the UPDATED codebook.  Passing the module's live buffer here
    (while ``codes`` is the pre-update snapshot) reproduces exactly that; with ``None`` the gradient is the consistent one.
    """
    x = x.float()
    if torch.is_grad_enabled() and (x.requires_grad or codes.requires_grad):
        return _SimilarityFn.apply(x, codes, metric, live_codes)
    return search.get_backend().similarities(x, codes, metric=metric)


def _rows_of(chunk):
    return chunk[0] if isinstance(chunk, tuple) else chunk


def _take(x, chunk):
    rows = _rows_of(chunk)
    return x[:, rows] if isinstance(rows, slice) else x.index_select(1, rows)


def _row_slices(m: int, step: int):
    return [slice(r, min(m, r + step)) for r in range(0, m, step)]


def _chunk_grads(chunk_value, x, codes, chunks, g, need_x, need_c):
    """d/d(x, codes) of  g * sum_chunks chunk_value(x[:, rows], codes, chunk), one chunk alive at a time."""
    x, codes = x.detach(), codes.detach()
    gx = torch.zeros_like(x) if need_x else None
    gc = torch.zeros_like(codes) if need_c else None
    for chunk in chunks:
        with torch.enable_grad():
            xc = _take(x, chunk).requires_grad_(need_x)
            cc = codes.requires_grad_(need_c) if need_c else codes
            wanted = [t for t, n in ((xc, need_x), (cc, need_c)) if n]
            grads = torch.autograd.grad(chunk_value(xc, cc, chunk), wanted, g)
        it = iter(grads)
        if need_x:
            rows = _rows_of(chunk)
            if isinstance(rows, slice):
                gx[:, rows] += next(it)
            else:
                gx.index_add_(1, rows, next(it))
        if need_c:
            gc += next(it)
    return gx, gc


class _ChunkedFn(torch.autograd.Function):
    """value = sum over row chunks of chunk_value(x[:, rows], codes, chunk);  recomputed chunk by chunk in backward, so
    only one [H, rows, K] chunk (and its temporaries) is alive at any time."""

    @staticmethod
    def forward(ctx, x, codes, chunk_value, chunks):
        total = None
        with torch.no_grad():
            for chunk in chunks:
                v = chunk_value(_take(x, chunk), codes, chunk)
                total = v if total is None else total + v
        ctx.save_for_backward(x, codes)
        ctx.chunk_value, ctx.chunks = chunk_value, chunks
        return total

    @staticmethod
    def backward(ctx, g):
        x, codes = ctx.saved_tensors
        gx, gc = _chunk_grads(ctx.chunk_value, x, codes, ctx.chunks, g, ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        return gx, gc, None, None


# ------------------------------------------------------------------------------------------------ cross entropy
class _CrossEntropyFn(torch.autograd.Function):
    """mean over non-ignored (h, m) of  logsumexp_k sims[h, m, k] - sims[h, m, target[h, m]]
    == F.cross_entropy(rearrange(distances, ...), codes, ignore_index=-1)   vector_quantize_pytorch.py:292-294.
    Forward: one native sweep (online softmax).  Backward: bounded chunks of the similarity matrix."""

    @staticmethod
    def forward(ctx, x, codes, target, metric, live_codes=None, lse=None, tl=None):
        ctx.live_codes = live_codes
        if lse is None:  # otherwise the search already produced them (vq_quantize_lse_f32): no second sweep
            lse, tl = search.get_backend().softmax_stats(x.detach(), codes.detach(), metric=metric, scale=1.0,
                                                         target=target)
        valid = target >= 0
        count = valid.sum()
        ctx.save_for_backward(x, codes, target, count, lse, tl)
        ctx.metric = metric
        return ((lse - tl) * valid).sum() / count

    @staticmethod
    def backward(ctx, g):
        x, codes, target, count, lse, tl = ctx.saved_tensors
        metric, live = ctx.metric, ctx.live_codes
        fused = getattr(search.get_backend(), "cross_entropy_backward", None)
        if fused is not None and live is None and not ctx.needs_input_grad[1]:
            # one fused sweep (S = x c^T, softmax weights, second contraction with the codebook) -- nothing of [M, K]
            coef = (g / count).reshape(1).to(torch.float32)
            gx = fused(x.detach(), codes.detach(), lse, tl, target, coef, metric=metric)
            if gx is not None:
                return gx, None, None, None, None, None, None

        def chunk_value(xc, cc, rows):
            sims = similarity_matrix(xc, cc, metric, live)
            return F.cross_entropy(sims.reshape(-1, sims.shape[-1]), target[:, rows].reshape(-1), ignore_index=-1,
                                   reduction="sum") / count

        chunks = _row_slices(x.shape[1], _rows_per_chunk(x.shape[0], codes.shape[1]))
        gx, gc = _chunk_grads(chunk_value, x, codes, chunks, g, ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        return gx, gc, None, None, None, None, None


def cross_entropy_to_codes(x: torch.Tensor, codes: torch.Tensor, target: torch.Tensor, metric: int,
                           live_codes=None, stats=None) -> torch.Tensor:
    """x [H, M, D] (strided rows fine), codes [H, K, D], target [H, M] int64 with -1 = ignore -> scalar.
    ``live_codes``: see similarity_matrix().  ``stats`` = (lse [H, M], target_logit [H, M]) when the search sweep already
    produced them for exactly this target (cross-entropy commitment loss: target = the chosen code)."""
    lse, tl = stats if stats is not None else (None, None)
    if lse is not None:
        lse, tl = lse.contiguous(), tl.contiguous()
    return _CrossEntropyFn.apply(x.float(), codes, target, metric, live_codes, lse, tl)


# ------------------------------------------------------------------------------------------------ diversity
def codebook_diversity_loss(x: torch.Tensor, codes: torch.Tensor, metric: int, temperature: float,
                            position: torch.Tensor, n_positions: int, live_codes=None) -> torch.Tensor:
    """-mean_n entropy(avg_prob[n]),  avg_prob[n] = mean over all (head, batch) rows at sequence position n of
    softmax(-similarities * temperature)   -- vector_quantize_pytorch.py:324-334 (sign and all).

    x [H, M, D]; position [M] int64 = sequence position of each row.  Chunked over positions: every chunk holds all
    rows of a set of positions, so its entropies are complete and the loss is a plain sum over chunks.
    """
    h, m, _ = x.shape
    k = codes.shape[1]
    counts = torch.bincount(position, minlength=n_positions)
    rows_per_pos = max(1, m // max(1, n_positions))
    pos_per_chunk = max(1, _rows_per_chunk(h, k) // rows_per_pos)
    if pos_per_chunk >= n_positions:
        chunks = [(slice(0, m), 0, n_positions)]
    else:
        order = torch.argsort(position, stable=True)  # rows grouped by position
        bounds = [0] + torch.cumsum(counts, 0).tolist()
        chunks = [(order[bounds[p0]:bounds[min(n_positions, p0 + pos_per_chunk)]], p0, min(n_positions, p0 + pos_per_chunk))
                  for p0 in range(0, n_positions, pos_per_chunk)]

    def chunk_value(xc, cc, chunk):
        rows, p0, p1 = chunk
        sims = similarity_matrix(xc, cc, metric, live_codes)
        prob = (-sims * temperature).softmax(dim=-1)  # [H, R, K]
        local = (position[rows] if isinstance(rows, slice) else position.index_select(0, rows)) - p0
        acc = torch.zeros((p1 - p0, k), dtype=prob.dtype, device=prob.device).index_add(0, local, prob.sum(0))
        avg = acc / (counts[p0:p1, None].to(prob.dtype) * h)
        ent = (-avg * avg.clamp(min=1e-5).log()).sum(-1)  # utils/general.py:25-30
        return -ent.sum() / n_positions

    return _ChunkedFn.apply(x.float(), codes, chunk_value, chunks)


# ------------------------------------------------------------------------------------------------ orthogonal
def orthogonal_loss(codes: torch.Tensor) -> torch.Tensor:
    """Eq. (2) of arXiv:2112.00384 on codes [h, n, d]  (utils/losses.py:23-28): mean squared cosine similarity
    between codes, minus 1/n.  Touches only the codebook, so it is a plain library GEMM."""
    h, n = codes.shape[:2]
    normed = F.normalize(codes, p=2, dim=-1)
    cos = normed @ normed.transpose(-1, -2)
    return (cos ** 2).sum() / (h * n ** 2) - (1 / n)
