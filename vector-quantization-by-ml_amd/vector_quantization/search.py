"""Functional seam between the nn.Modules and the native op.

``quantize_rows`` is the ONE call every module makes for the hot path.  It dispatches to
``native.quantize`` (HIP kernels through the C ABI).  There is no CPU implementation in this package -- not for the
search, not for the similarity consumers, not for the EMA statistics: every device-side step goes through the backend
object below.  Tests may install a checker backend with ``set_backend`` (tests/ only) to exercise the host-side
layout logic on a machine without a GPU.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import native

EUCLID = native.EUCLID
DOT = native.DOT


LSE_MAX_DIM = 512  # vq_quantize_lse_f32: rows of one launch


def _want_deterministic(dim: int) -> bool:
    """torch.use_deterministic_algorithms(True) selects the reproducible EMA accumulation (rows of up to 2048 dims).
    Wider rows only have the float-atomic kernel: torch's own convention applies -- raise, or warn under ``warn_only``."""
    if not torch.are_deterministic_algorithms_enabled():
        return False
    if dim <= 2048:
        return True
    msg = (f"the EMA statistics of rows wider than 2048 dims (got {dim}) are accumulated with float atomics and have no "
           "deterministic implementation; turn torch.use_deterministic_algorithms off (or use warn_only=True) for this step")
    if torch.is_deterministic_algorithms_warn_only_enabled():
        import warnings

        warnings.warn(msg)
        return False
    raise RuntimeError(msg)


class _NativeBackend:
    name = "hip-gfx950"
    accepts_half_rows = True  # fp16 / bf16 rows are widened in the kernel's prologue (inference launches)
    uses_packed = True        # callers may hand over a cached packed image of the codebooks (``pack``)

    @staticmethod
    @torch.compiler.assume_constant_result  # host arithmetic in the library: a constant of (dim, want_sq_err) for a trace
    def max_fused_stages(dim, want_sq_err):
        """How many residual stages one launch can hold (vq_max_fused_stages); longer stacks run layer by layer."""
        return native.max_fused_stages(dim, want_sq_err)

    @staticmethod
    def pack(cb, metric):
        """cb [..., K, D] contiguous fp32 -> packed images [n, packed_floats] (vq_pack_codebooks_f32)."""
        if torch.compiler.is_compiling():
            from . import ops

            return ops.pack(cb, metric)
        return native.pack_codebooks(cb, metric)

    @staticmethod
    def quantize(x, cb, *, metric, ste, want_sq_err, share, want_best=False, out=None, idx=None, want_lse=False,
                 sq_err_per_head=False, packed=None):
        """-> (out, idx, best, sq_err) and, with ``want_lse`` (single stage), a fifth element: the per-row log-sum-exp
        of the similarities from the same sweep (vq_quantize_lse_f32)."""
        if torch.compiler.is_compiling() and not (want_best or want_lse):
            # being traced by torch.compile / export: go through the registered op (ops.py) so the graph does not break here
            from . import ops

            H, M, D = x.shape
            Q = idx.shape[-1] if idx is not None else (1 if share else cb.shape[1])
            if out is None:
                out = torch.empty((H, M, D), dtype=torch.float32, device=x.device)
            if idx is None:
                idx = torch.empty((H, M, Q), dtype=torch.int64, device=x.device)
            sq_err = ops.quantize_into(x, cb, packed, out, idx, metric, ste, want_sq_err, share, sq_err_per_head)
            return out, idx, None, (sq_err if want_sq_err else None)
        r = native.quantize(x, cb, metric=metric, ste=ste, want_sq_err=want_sq_err, want_best=want_best or want_lse,
                            stages_share_codebook=share, out=out, idx=idx, want_lse=want_lse,
                            sq_err_per_head=sq_err_per_head, packed=packed)
        if want_lse:
            return r["out"], r["idx"], r["best"], r["sq_err"], r["lse"]
        return r["out"], r["idx"], r["best"], r["sq_err"]

    @staticmethod
    def shard_keys(x, cb, *, metric, idx_offset, packed=None):
        """Search ONE SHARD of a codebook: x [H, M, D], cb [H, K_local, D] -> packed signed 64-bit keys, candidate planes
        [P, H, M] ((order image of the winning value) << 32 | idx_offset + local index; vq_search_key_planes_f32: one
        launch, no init, no atomics -- P = the K splits that fill the chip for this shape).  The element-wise MIN over the
        planes of all shards is the whole codebook's winner, lowest index on ties."""
        return native.search_key_planes(x, cb, metric=metric, idx_offset=idx_offset, packed=packed)

    @staticmethod
    def finalize_keys(x, table, keys, *, metric):
        """Keys [H, M] or candidate planes [C, H, M] (MIN taken on the fly) + a FULL natural table [H, K, D] ->
        (quantized rows [H, M, D], idx [H, M]) (vq_finalize_key_planes_f32)."""
        r = native.finalize_keys(x, table, keys, metric=metric)
        return r["out"], r["idx"]

    @staticmethod
    def similarities(x, cb, *, metric, out=None):
        """x [H, M, D], cb [H, K, D] -> [H, M, K]  (vq_similarities_f32)."""
        return native.similarities(x, cb.contiguous(), metric=metric, out=out)

    @staticmethod
    def softmax_stats(x, cb, *, metric, scale, target=None):
        """-> (logsumexp_k scale * sim [H, M], logit of target [H, M] | None)  (vq_softmax_stats_f32)."""
        return native.softmax_stats(x, cb.contiguous(), metric=metric, scale=scale, target=target)


    @staticmethod
    def ema_accumulate(x, idx, k, mask=None):
        """-> (counts [H, K], sums [H, K, D]) of the rows assigned to each code (vq_ema_accumulate_f32; under
        torch.use_deterministic_algorithms(True) the atomics-free vq_ema_accumulate_det_f32)."""
        return native.ema_accumulate(x, idx, k, mask, deterministic=_want_deterministic(x.shape[-1]))

    @staticmethod
    def ema_accumulate_residual(x, cb, idx, *, ste, share):
        """-> (counts [H, Q, K], sums [H, Q, K, D]) for every stage of a residual stack (vq_ema_accumulate_residual_f32)."""
        return native.ema_accumulate_residual(x, cb, idx, ste=ste, stages_share_codebook=share,
                                              deterministic=_want_deterministic(x.shape[-1]))

    @staticmethod
    def ema_update(cluster_size, embed_avg, embeddings, hits, sums, *, decay, eps, l2norm):
        """In-place lerp / Laplace smoothing / normalise of the module buffers (vq_ema_update_f32)."""
        native.ema_update(cluster_size, embed_avg, embeddings, hits, sums, decay, eps, l2norm)

    @staticmethod
    def quantize_backward(x, cb, idx, grad_out, grad_sq_err, *, ste, share, sq_err_per_head=False):
        """d/dx of the quantize step in one native pass (vq_quantize_backward_f32)."""
        return native.quantize_backward(x, cb, idx, grad_out, grad_sq_err, ste=ste, stages_share_codebook=share,
                                        sq_err_per_head=sq_err_per_head)

    @staticmethod
    def cross_entropy_backward(x, cb, lse, target_logit, target, coef, *, metric):
        """Fused d/dx of the cross entropy (vq_ce_backward_f32); None when the shape is outside the kernel's range."""
        if x.shape[-1] > native.CE_BACKWARD_MAX_DIM:
            return None
        return native.ce_backward(x, cb.contiguous(), lse, target_logit, target, coef, metric=metric)


_backend = _NativeBackend


def set_backend(backend) -> None:
    """Install a different backend object exposing ``quantize`` / ``similarities`` / ``softmax_stats`` (used by the
    CPU-only host-logic tests)."""
    global _backend
    _backend = backend if backend is not None else _NativeBackend


def get_backend():
    return _backend


class _QuantizeFn(torch.autograd.Function):
    """Autograd wrapper.  The native op has no backward of its own; the gradients are the reference's:

    * train (ste): out_q = r_q + (c_q - r_q).detach()  -> d out / d x = Q * I   (residual_vq.py:232-233,
      vector_quantize_pytorch.py:273: every stage's residual is x minus detached terms)
    * sq_err_q = sum (c_q.detach() - r_q)^2              -> d / d x = 2 (r_q - c_q)
    * a learnable codebook receives  2 (c_q - r_q)  from sq_err when ``codebook_grad_from_err``
      (commitment loss not detached, vector_quantize_pytorch.py:263-269) and, in eval, the scatter of
      grad_out (out = codebook[idx]).
    """

    @staticmethod
    def forward(ctx, x, cb, metric, ste, want_sq_err, share, codebook_grad_from_err, out_buf, idx_buf, want_lse=False,
                per_head=False, packed=None):
        extra = {}
        if want_lse:
            extra["want_lse"] = True
        if per_head:
            extra["sq_err_per_head"] = True
        if packed is not None:
            extra["packed"] = packed
        res = _backend.quantize(x.detach(), cb.detach(), metric=metric, ste=ste, want_sq_err=want_sq_err, share=share,
                                out=out_buf, idx=idx_buf, **extra)
        out, idx, best, sq_err = res[:4]
        ctx.save_for_backward(x, cb, idx)
        ctx.ste, ctx.share, ctx.cb_err, ctx.per_head = ste, share, codebook_grad_from_err, per_head
        if sq_err is None:
            sq_err = torch.zeros((x.shape[0], idx.shape[-1]) if per_head else idx.shape[-1], dtype=torch.float64,
                                 device=x.device)
        if want_lse:
            ctx.mark_non_differentiable(idx, best, res[4])
            return out, idx, sq_err, best, res[4]
        ctx.mark_non_differentiable(idx)
        return out, idx, sq_err

    @staticmethod
    def backward(ctx, g_out, _g_idx, g_err, *_unused):
        x, cb, idx = ctx.saved_tensors
        H, M, D = x.shape
        Q = idx.shape[-1]
        gx = gcb = None
        need_x = ctx.needs_input_grad[0]
        need_cb = ctx.needs_input_grad[1]
        fused = getattr(_backend, "quantize_backward", None)
        if fused is not None and need_x and not need_cb:
            # one pass over x / grad_out; no host synchronisation (the torch path below inspects g_err on the host)
            return (fused(x.detach(), cb.detach(), idx, g_out if ctx.ste else None, g_err, ste=ctx.ste, share=ctx.share,
                          **({"sq_err_per_head": True} if ctx.per_head else {})),
                    None, None, None, None, None, None, None, None, None, None, None)
        if need_x:
            gx = g_out * float(Q) if ctx.ste else torch.zeros_like(x)
        if need_cb:
            gcb = torch.zeros_like(cb)
        has_err = g_err is not None and bool((g_err != 0).any())
        if (need_x and has_err) or need_cb:
            r = x.detach()
            harange = torch.arange(H, device=x.device)[:, None]
            for q in range(Q):
                cq = cb[:, 0 if ctx.share else q].detach()  # [H, K, D]
                i = idx[..., q]
                c = cq[harange, i]  # [H, M, D]
                if has_err:
                    w = (g_err[:, q, None, None] if ctx.per_head else g_err[q]).to(x.dtype)
                    if need_x:
                        gx = gx + 2.0 * w * (r - c)
                    if need_cb and ctx.cb_err:
                        gcb[:, 0 if ctx.share else q].index_put_((harange.expand_as(i), i), 2.0 * w * (c - r),
                                                                 accumulate=True)
                if need_cb and not ctx.ste and g_out is not None:
                    gcb[:, 0 if ctx.share else q].index_put_((harange.expand_as(i), i), g_out, accumulate=True)
                quant = r + (c - r) if ctx.ste else c
                r = r - quant
        return gx, gcb, None, None, None, None, None, None, None, None, None, None


def quantize_rows(x: torch.Tensor, cb: torch.Tensor, *, metric: int = EUCLID, ste: bool = False,
                  want_sq_err: bool = False, share: bool = False, codebook_grad_from_err: bool = False,
                  out: Optional[torch.Tensor] = None, idx: Optional[torch.Tensor] = None, want_lse: bool = False,
                  sq_err_per_head: bool = False, packed: Optional[torch.Tensor] = None):
    """x [H, M, D] (strided rows allowed), cb [H, Q, K, D] contiguous ([H, 1, K, D] when ``share``; the number
    of stages is then ``idx.shape[-1]``).  ``out`` / ``idx`` may be pre-allocated (strided) destination views.

    Returns (out [H, M, D], idx [H, M, Q] int64, sq_err [Q] float64 or None); with ``want_lse`` (single stage) a fourth
    element: dict(best [H, M, 1], lse [H, M]) -- the winner's distance / similarity and the log-sum-exp of the row's
    similarities, both from the same sweep (what the cross-entropy commitment loss needs).
    ``sq_err_per_head``: sq_err is [H, Q] (one sum per head: GroupedResidualVQ reports a loss per group).
    ``packed``: a cached packed image of ``cb`` from ``get_backend().pack`` (backends with ``uses_packed``); without it
    the native backend packs on every call.
    """
    if packed is not None and not getattr(_backend, "uses_packed", False):
        packed = None
    if not cb.is_contiguous():
        cb = cb.contiguous()
    # x receives a gradient only through the straight-through output or the squared error: an eval-mode gather
    # (out = codebook[idx]) is not differentiable with respect to x, exactly like the reference's batched_embedding
    needs_grad = torch.is_grad_enabled() and ((x.requires_grad and (ste or want_sq_err)) or cb.requires_grad)
    if x.dtype != torch.float32:
        half_rows = (x.dtype in (torch.float16, torch.bfloat16) and getattr(_backend, "accepts_half_rows", False)
                     and not needs_grad and not ste and not want_sq_err and not want_lse and cb.shape[1] == 1
                     and (idx is None or idx.shape[-1] == 1))
        if not half_rows:
            x = x.float()  # the reference's x.float() (codebooks.py:354); the native inference launch does it in-kernel
    if needs_grad:
        res = _QuantizeFn.apply(x, cb, metric, ste, want_sq_err, share, codebook_grad_from_err, out, idx, want_lse,
                                sq_err_per_head, packed)
        out, idx, sq_err = res[:3]
        if want_lse:
            return out, idx, (sq_err if want_sq_err else None), dict(best=res[3], lse=res[4])
        return out, idx, (sq_err if want_sq_err else None)
    pk = {"packed": packed} if packed is not None else {}
    if want_lse:
        out, idx, best, sq_err, lse = _backend.quantize(x, cb, metric=metric, ste=ste, want_sq_err=want_sq_err,
                                                        share=share, out=out, idx=idx, want_lse=True, **pk)
        return out, idx, sq_err, dict(best=best, lse=lse)
    out, idx, _best, sq_err = _backend.quantize(x, cb, metric=metric, ste=ste, want_sq_err=want_sq_err, share=share,
                                                out=out, idx=idx, **pk,
                                                **({"sq_err_per_head": True} if sq_err_per_head else {}))
    return out, idx, sq_err


def nearest_with_distance(x: torch.Tensor, cb: torch.Tensor, *, metric: int = EUCLID):
    """Search only: (idx [H, M], best [H, M]) -- used by k-means seeding and diagnostics."""
    out, idx, best, _ = _backend.quantize(x.float(), cb[:, None].contiguous(), metric=metric, ste=False,
                                          want_sq_err=False, share=False, want_best=True)
    return idx[..., 0], best[..., 0], out
