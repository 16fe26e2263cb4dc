"""``ResidualVQ`` / ``GroupedResidualVQ`` (reference: /root/reference/vector_quantization/residual_vq.py:26-357).

The reference walks its layers in Python, and every layer re-reads / re-writes the full residual and
rebuilds the ``[rows, K]`` similarity and one-hot tensors (residual_vq.py:212-243).  Here the whole stack
of stages is ONE native launch when the layers are homogeneous (the normal case): the residual rows stay
in registers across the Q codebook sweeps, the kernel writes ``indices[..., q]``, the accumulated
``quantized_out`` and the per-stage squared errors.  ``GroupedResidualVQ`` maps its groups onto the
kernel's head axis, so G groups x Q stages are still one launch.

Heterogeneous stacks (masks, per-layer input normalisation, quantize-dropout, k-means seeding on the
first batch, channel-first layers) fall back to a per-layer loop in which each layer call is itself a
native launch.
"""
from __future__ import annotations

import random
from typing import List, Optional

import torch
import torch.distributed as dist
from torch import nn

from . import search
from .quantizer import VectorQuantize, _cached_zeros


def _round_up(value: int, multiple: int) -> int:
    return -(-value // multiple) * multiple


def _residual_chain(x: torch.Tensor, codes: torch.Tensor, idx: torch.Tensor, ste: bool) -> List[torch.Tensor]:
    """Per-stage residuals [r_1 .. r_Q] recomputed from the indices with the reference's fp32 operations
    (used only for training-state bookkeeping after the fused launch).  x [M, D], codes [Q, K, D], idx [M, Q]."""
    residuals = []
    r = x
    for q in range(idx.shape[-1]):
        residuals.append(r)
        c = codes[q][idx[:, q]]
        quant = r + (c - r) if ste else c
        r = r - quant
    return residuals


class ResidualVQ(nn.Module):
    """Follows Algorithm 1 of https://arxiv.org/abs/2107.03312 (SoundStream), like the reference."""

    def __init__(
        self,
        *,
        dim,
        num_quantizers,
        codebook_dim=None,
        shared_codebook=False,
        heads=1,
        quantize_dropout=False,
        quantize_dropout_cutoff_index=0,
        quantize_dropout_multiple_of=1,
        **kwargs,
    ):
        super().__init__()
        assert heads == 1, "residual vq is not compatible with multi-headed codes"
        inner = dim if codebook_dim is None else codebook_dim
        self.has_projections = inner != dim
        self.project_in = nn.Linear(dim, inner) if self.has_projections else nn.Identity()
        self.project_out = nn.Linear(inner, dim) if self.has_projections else nn.Identity()
        self.num_quantizers = num_quantizers
        self.layers = nn.ModuleList(
            [VectorQuantize(dim=inner, codebook_dim=inner, **kwargs) for _ in range(num_quantizers)]
        )
        assert all(not layer.has_projections for layer in self.layers)

        self.quantize_dropout = quantize_dropout and num_quantizers > 1
        assert quantize_dropout_cutoff_index >= 0
        self.quantize_dropout_cutoff_index = quantize_dropout_cutoff_index
        self.quantize_dropout_multiple_of = quantize_dropout_multiple_of
        self.shared_codebook = shared_codebook
        if shared_codebook:
            first = self.layers[0]._codebook
            for layer in self.layers[1:]:
                layer._codebook = first
        self._stage_cache = None   # (key, stacked natural codebooks [1, Q|1, K, D], packed images) of the fused launch
        self._zero_losses = None

    # ------------------------------------------------------------------ codes
    @property
    def codebooks(self):
        """[Q, K, D] stack of the per-layer codebooks."""
        return torch.stack([layer._codebook.embeddings[0] for layer in self.layers], dim=0)

    def get_codes_from_indices(self, indices):
        """indices [b, ..., q] (q may be < Q, -1 = dropped) -> [Q, b, ..., D]."""
        q_given = indices.shape[-1]
        if q_given < self.num_quantizers:
            assert self.quantize_dropout > 0.0, (
                "quantize dropout must be greater than 0 if you wish to reconstruct from a signal with less fine "
                "quantizations"
            )
            pad = indices.new_full((*indices.shape[:-1], self.num_quantizers - q_given), -1)
            indices = torch.cat([indices, pad], dim=-1)
        dropped = indices < 0
        safe = indices.clamp(min=0)
        books = self.codebooks
        per_stage = [books[q][safe[..., q]] for q in range(self.num_quantizers)]
        codes = torch.stack(per_stage, dim=0)
        return codes.masked_fill(dropped.movedim(-1, 0)[..., None], 0.0)

    def get_output_from_indices(self, indices):
        return self.project_out(self.get_codes_from_indices(indices).sum(dim=0))

    # ------------------------------------------------------------------ forward
    def _fusable(self, x, mask, drop_active: bool) -> bool:
        if mask is not None or drop_active:
            return False
        first = self.layers[0]
        cb0 = first._codebook
        # the kernel keeps every stage's winners (and loss partials) in LDS: stacks beyond its budget (e.g. 32 stages of
        # dim 256 with the commitment loss) run layer by layer, one launch each, like the reference's loop
        limit = getattr(search.get_backend(), "max_fused_stages", None)
        if limit is not None and self.num_quantizers > limit(cb0.dim, self.training and first.has_commitment_loss):
            return False
        for layer in self.layers:
            cb = layer._codebook
            if (not layer.channel_last or not cb.is_initialized or cb._stochastic_requested()
                    or cb.transform_input is not cb0.transform_input
                    or cb.transform_input.__name__ != "_identity"
                    or cb.metric != cb0.metric or cb.embeddings.shape != cb0.embeddings.shape
                    or layer.commitment_weight != first.commitment_weight
                    or cb.learnable_codebook
                    # losses that look at the similarities go through VectorQuantize.forward layer by layer
                    or layer.commitment_use_cross_entropy_loss or layer.has_codebook_diversity_loss
                    or layer.has_codebook_orthogonal_loss):
                return False
        return True

    def forward(self, x, mask=None, indices=None, return_all_codes=False, freeze_codebook=False,
                rand_quantize_dropout_fixed_seed=None):
        assert indices is None, "cross-entropy to given indices is not supported (asserted in the reference as well)"
        x = self.project_in(x)
        drop_active = self.training and self.quantize_dropout
        # a SHARED codebook that is being updated must be searched stage by stage: in the reference every layer's forward
        # rewrites it (EMA) before the next layer looks at it (residual_vq.py:212-233, codebooks.py:399-426)
        shared_and_moving = (self.shared_codebook and self.training and not freeze_codebook
                             and self.layers[0]._codebook.ema_update)
        if not shared_and_moving and self._fusable(x, mask, drop_active):
            quantized, all_indices, all_losses = self._forward_fused(x, freeze_codebook)
        else:
            quantized, all_indices, all_losses = self._forward_layers(x, mask, freeze_codebook, drop_active,
                                                                      rand_quantize_dropout_fixed_seed)
        quantized = self.project_out(quantized)
        ret = (quantized, all_indices, all_losses)
        if return_all_codes:
            ret = (*ret, self.get_codes_from_indices(all_indices))
        return ret

    def _stage_codes(self):
        """Stacked natural codebooks [1, Q, K, D] ([1, 1, K, D] when shared) and their packed images for the fused launch,
        rebuilt only when some layer's codes changed (Codebook.codes_state): an inference forward neither re-stacks nor
        re-packs."""
        cbs = [self.layers[0]._codebook] if self.shared_codebook else [layer._codebook for layer in self.layers]
        backend = search.get_backend()
        if torch.compiler.is_compiling():  # traced: stack and pack are nodes of the graph
            codes = torch.stack([cb.embeddings.detach()[0] for cb in cbs], dim=0)[None].contiguous()
            return codes, (backend.pack(codes, cbs[0].metric) if getattr(backend, "uses_packed", False) else None)
        key = (tuple(cb.codes_state() for cb in cbs), cbs[0].metric, getattr(backend, "name", None))
        if self._stage_cache is None or self._stage_cache[0] != key:
            with torch.no_grad():
                codes = torch.stack([cb.embeddings.detach()[0] for cb in cbs], dim=0)[None].contiguous()
                packed = None
                if getattr(backend, "uses_packed", False) and codes.is_cuda:
                    packed = backend.pack(codes, cbs[0].metric)
            self._stage_cache = (key, codes, packed)
        return self._stage_cache[1], self._stage_cache[2]

    def _forward_fused(self, x, freeze_codebook):
        first = self.layers[0]
        cb0 = first._codebook
        training = self.training
        lead, d = x.shape[:-1], x.shape[-1]
        flat = x.reshape(1, x.numel() // max(d, 1), d)
        wide_input = flat.dtype == torch.float64  # train mode: the reference's straight-through sums are in the input's width
        if flat.dtype != torch.float32:
            flat = flat.float()
        Q = self.num_quantizers
        codes, packed = self._stage_codes()  # [1, Q, K, D] ([1, 1, K, D] when shared); fusable stacks are not learnable
        want_loss = training and first.has_commitment_loss
        idx_buf = torch.empty((1, flat.shape[1], Q), dtype=torch.int64, device=flat.device)
        out, idx, sq_err = search.quantize_rows(flat, codes, metric=cb0.metric, ste=training, want_sq_err=want_loss,
                                                share=self.shared_codebook, idx=idx_buf, packed=packed)
        if want_loss:
            losses = (sq_err / flat.numel()).to(torch.float32)[None, :] * first.commitment_weight
        elif training:
            losses = torch.zeros((1, Q), dtype=torch.float32, device=flat.device)
        else:
            losses = _cached_zeros(self, "_zero_losses", (1, Q), flat.device)  # no fill kernel per inference forward

        if training and not freeze_codebook and cb0.ema_update:
            with torch.no_grad():
                stage_codes = codes[0].expand(Q, -1, -1) if self.shared_codebook else codes[0]
                chain = []

                def residual_rows(q):
                    if not chain:  # only needed when a code expired
                        chain.extend(_residual_chain(flat[0].detach(), stage_codes.detach(), idx[0], ste=True))
                    return chain[q][None]

                # one pass rebuilds the residual chain from the indices and scatter-adds every stage's statistics
                hits, sums = search.get_backend().ema_accumulate_residual(flat.detach(), codes.detach().contiguous(), idx,
                                                                          ste=True, share=self.shared_codebook)
                for q, layer in enumerate(self.layers):
                    layer._codebook.ema_apply(hits[:, q], sums[:, q])
                    layer._codebook.reseed_dead_codes(lambda q=q: residual_rows(q))
        if training and wide_input:
            out = out.double()
        return out.reshape(*lead, d), idx.reshape(*lead, Q), losses

    def _forward_layers(self, x, mask, freeze_codebook, drop_active, fixed_seed):
        Q = self.num_quantizers
        cut = Q
        if drop_active:
            if fixed_seed is not None:
                rng = random.Random(fixed_seed)
            elif dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
                seed = torch.tensor(random.randrange(10_000), device=x.device)
                dist.all_reduce(seed)
                rng = random.Random(int(seed.item()))
            else:
                rng = random
            cut = rng.randrange(self.quantize_dropout_cutoff_index, Q)
            if self.quantize_dropout_multiple_of != 1:
                cut = _round_up(cut + 1, self.quantize_dropout_multiple_of) - 1
        residual = x
        quantized_out = 0.0
        all_indices, all_losses = [], []
        for q, layer in enumerate(self.layers):
            if drop_active and q > cut:
                shape = (x.shape[0], *x.shape[-2:]) if x.ndim >= 4 else tuple(x.shape[:2])
                all_indices.append(torch.full(shape, -1, device=x.device, dtype=torch.long))
                all_losses.append(torch.zeros(1, device=x.device, dtype=x.dtype))
                continue
            quantized, layer_idx, layer_loss = layer(residual, mask=mask, freeze_codebook=freeze_codebook)
            residual = residual - quantized.detach()
            quantized_out = quantized_out + quantized
            all_indices.append(layer_idx)
            all_losses.append(layer_loss)
        return quantized_out, torch.stack(all_indices, dim=-1), torch.stack(all_losses, dim=-1)


class GroupedResidualVQ(nn.Module):
    def __init__(self, *, dim, groups=1, channel_last=True, **kwargs):
        super().__init__()
        assert dim % groups == 0
        self.dim = dim
        self.groups = groups
        self.channel_last = channel_last
        self.rvqs = nn.ModuleList([ResidualVQ(dim=dim // groups, **kwargs) for _ in range(groups)])
        self._stage_cache = None
        self._zero_losses = None

    @property
    def split_dim(self) -> int:
        return -1 if self.channel_last else 1

    @property
    def codebooks(self):
        return torch.stack(tuple(rvq.codebooks for rvq in self.rvqs))

    def get_codes_from_indices(self, indices):
        return torch.stack(tuple(rvq.get_codes_from_indices(i) for rvq, i in zip(self.rvqs, indices)))

    def get_output_from_indices(self, indices):
        outs = tuple(rvq.get_output_from_indices(i) for rvq, i in zip(self.rvqs, indices))
        return torch.cat(outs, dim=self.split_dim)

    def _fusable(self, x, mask) -> bool:
        if not self.channel_last or mask is not None:
            return False
        r0 = self.rvqs[0]
        for rvq in self.rvqs:
            if rvq.has_projections or rvq.shared_codebook or rvq.num_quantizers != r0.num_quantizers:
                return False
            if not rvq._fusable(x, None, rvq.training and rvq.quantize_dropout):
                return False
            if rvq.layers[0]._codebook.embeddings.shape != r0.layers[0]._codebook.embeddings.shape:
                return False
            if rvq.layers[0]._codebook.metric != r0.layers[0]._codebook.metric:
                return False
        return True

    def forward(self, x, indices=None, return_all_codes=False, freeze_codebook=False, mask=None):
        assert x.shape[self.split_dim] == self.dim
        assert indices is None or len(indices) == 0, "cross-entropy to given indices is not supported"
        if self._fusable(x, mask):
            return self._forward_fused(x, return_all_codes, freeze_codebook)
        chunks = x.chunk(self.groups, dim=self.split_dim)
        seed = random.randint(0, int(1e7))
        outs = tuple(
            rvq(chunk, mask=mask, return_all_codes=return_all_codes, freeze_codebook=freeze_codebook,
                rand_quantize_dropout_fixed_seed=seed)
            for rvq, chunk in zip(self.rvqs, chunks)
        )
        quantized, all_indices, losses, *maybe_codes = tuple(zip(*outs))
        ret = (torch.cat(quantized, dim=self.split_dim), torch.stack(all_indices), torch.stack(losses))
        if maybe_codes:
            ret = (*ret, torch.stack(maybe_codes[0]))
        return ret

    def _stage_codes(self):
        """[G, Q, K, d] natural codebooks + packed images of the one fused launch, rebuilt only when some codes changed."""
        cbs = [layer._codebook for rvq in self.rvqs for layer in rvq.layers]
        backend = search.get_backend()
        if torch.compiler.is_compiling():  # traced: stack and pack are nodes of the graph (no identity-keyed cache in a trace)
            codes = torch.stack([torch.stack([layer._codebook.embeddings.detach()[0] for layer in rvq.layers], dim=0)
                                 for rvq in self.rvqs], dim=0).contiguous()
            return codes, (backend.pack(codes, cbs[0].metric) if getattr(backend, "uses_packed", False) else None)
        key = (tuple(cb.codes_state() for cb in cbs), cbs[0].metric, getattr(backend, "name", None))
        if self._stage_cache is None or self._stage_cache[0] != key:
            with torch.no_grad():
                codes = torch.stack([torch.stack([layer._codebook.embeddings.detach()[0] for layer in rvq.layers], dim=0)
                                     for rvq in self.rvqs], dim=0).contiguous()
                packed = None
                if getattr(backend, "uses_packed", False) and codes.is_cuda:
                    packed = backend.pack(codes, cbs[0].metric)
            self._stage_cache = (key, codes, packed)
        return self._stage_cache[1], self._stage_cache[2]

    def _forward_fused(self, x, return_all_codes, freeze_codebook):
        """Groups on the kernel's head axis: x [..., G*d] is searched in place as a [G, rows, d] view."""
        G = self.groups
        r0 = self.rvqs[0]
        Q = r0.num_quantizers
        first = r0.layers[0]
        cb0 = first._codebook
        training = self.training
        lead = x.shape[:-1]
        d = self.dim // G
        xc = x if (x.is_contiguous() and x.dtype == torch.float32) else x.contiguous().float()
        rows = xc.numel() // self.dim
        flat = xc.view(rows, G, d).permute(1, 0, 2)
        codes, packed = self._stage_codes()  # [G, Q, K, d]
        want_loss = training and first.has_commitment_loss
        q_buf = torch.empty((rows, G, d), dtype=torch.float32, device=xc.device)
        # one launch for all groups and stages; squared errors come back per group (head) and stage
        out, idx, sq_err = search.quantize_rows(flat, codes, metric=cb0.metric, ste=training, want_sq_err=want_loss,
                                                out=q_buf.permute(1, 0, 2), sq_err_per_head=True, packed=packed)
        if want_loss:
            losses = (sq_err / (rows * d)).to(torch.float32)[:, None, :] * first.commitment_weight
        elif training:
            losses = torch.zeros((G, 1, Q), dtype=torch.float32, device=xc.device)
        else:
            losses = _cached_zeros(self, "_zero_losses", (G, 1, Q), xc.device)
        if training and not freeze_codebook and cb0.ema_update:
            with torch.no_grad():
                chains = {}

                def residual_rows(g, q):
                    if g not in chains:  # only when a code expired
                        chains[g] = _residual_chain(flat[g].detach(), codes[g].detach(), idx[g], ste=True)
                    return chains[g][q][None]

                hits, sums = search.get_backend().ema_accumulate_residual(flat.detach(), codes.detach().contiguous(), idx,
                                                                          ste=True, share=False)
                for g, rvq in enumerate(self.rvqs):
                    for q, layer in enumerate(rvq.layers):
                        cbq = layer._codebook
                        if cbq.ema_update:
                            cbq.ema_apply(hits[g:g + 1, q], sums[g:g + 1, q])
                            cbq.reseed_dead_codes(lambda g=g, q=q: residual_rows(g, q))
        quantized = out.permute(1, 0, 2).reshape(*lead, self.dim)  # q_buf's memory; keeps the autograd edge of `out`
        if training and x.dtype == torch.float64:
            quantized = quantized.double()
        all_indices = idx.reshape(G, *lead, Q)
        ret = (quantized, all_indices, losses)
        if return_all_codes:
            ret = (*ret, self.get_codes_from_indices(all_indices))
        return ret
