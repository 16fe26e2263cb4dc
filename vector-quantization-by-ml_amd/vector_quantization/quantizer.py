"""``VectorQuantize``: public nn.Module of the drop-in (reference:
/root/reference/vector_quantization/vector_quantize_pytorch.py:38-430).

Same constructor / ``forward`` signature, sub-module names (``_codebook``, ``project_in``, ``project_out``)
and return convention ``(quantize, embed_ind, loss)``.  The body is organised differently from the
reference: every input layout is reduced to a *strided view* ``[heads, rows, dim]`` of one buffer, the
whole quantize step -- search, gather, straight-through output, squared error for the commitment
loss -- is ONE native call on that view (no head transposes, no ``[rows, K]`` intermediates), and the
results are viewed back.
"""
from __future__ import annotations

from collections import namedtuple
from dataclasses import asdict, replace
from typing import Callable, Optional

import torch
import torch.distributed as dist
from torch import nn

from . import losses
from .codebook import Codebook
from .params import CodebookParams

LossBreakdown = namedtuple(
    "LossBreakdown", ["commitment", "codebook_diversity", "orthogonal_reg", "inplace_optimize"]
)


def _cached_zeros(module, attr: str, shape, device) -> torch.Tensor:
    """A persistent all-zero fp32 tensor for the loss an inference forward returns (the reference allocates one per call,
    i.e. a fill kernel per forward).  Re-created if the caller modified the previous one in place; successive inference
    forwards return the SAME tensor object until then.  The tensor is created outside inference mode so that it keeps a
    version counter (an inference tensor has none: ``torch.inference_mode()`` is the standard serving idiom)."""
    if torch.compiler.is_compiling():
        return torch.zeros(shape, dtype=torch.float32, device=device)
    entry = getattr(module, attr, None)
    if (entry is None or entry[0].device != device or tuple(entry[0].shape) != tuple(shape)
            or entry[0]._version != entry[1]):
        with torch.inference_mode(False), torch.no_grad():
            t = torch.zeros(shape, dtype=torch.float32, device=device)
        entry = (t, t._version)
        setattr(module, attr, entry)
    return entry[0]


def _stochastic_sampling_requested(params: CodebookParams) -> bool:
    """Gumbel-max code sampling asked for (the rule of Codebook._stochastic_requested, on the config before it is built)."""
    g = params.gumbel_params
    g = asdict(g) if hasattr(g, "__dataclass_fields__") else dict(g or {})
    return bool(g.get("training", True) and g.get("stochastic", False) and g.get("temperature", 1.0) > 0)


def _world_is_distributed() -> bool:
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


class VectorQuantize(nn.Module):
    def __init__(
        self,
        dim,
        codebook_params: CodebookParams,
        codebook_dim=None,
        heads=1,
        separate_codebook_per_head=False,
        layernorm_after_project_in=False,
        channel_last=True,
        commitment_weight=1.0,
        commitment_use_cross_entropy_loss=False,
        orthogonal_reg_weight=0.0,
        orthogonal_reg_active_codes_only=False,
        orthogonal_reg_max_codes=None,
        codebook_diversity_loss_weight=0.0,
        codebook_diversity_temperature=100.0,
        sync_codebook=None,
        in_place_codebook_optimizer: Optional[Callable] = None,
        sync_update_v=0.0,
        codebook_shard_group=None,
        codebook_shard_reduction: str = "all_gather",
        codebook_shard_gather: str = "owner",
    ):
        """Arguments up to ``sync_update_v`` are the reference's (vector_quantize_pytorch.py:39-60).  New, keyword-only in
        spirit: ``codebook_shard_group`` (a process group, or True for the default group) shards the codebook's K codes over
        the ranks -- ``codebook_params.codebook_size`` stays the GLOBAL K, ``_codebook`` (and the checkpoint) hold this
        rank's ``[K / G, D]`` rows, every rank sees the same tokens, and the forward is shard-local search -> packed
        (distance, index) keys -> one collective (``codebook_shard_reduction``: "all_gather" + local min, one hop over all
        xGMI links, or "all_reduce" with MIN) -> gather of the winning rows from their owners
        (``codebook_shard_gather``: "owner" = SUM all-reduce of the owners' rows, no replicated table; "replicated" = an
        all-gathered full table cached on every rank).  BASELINE configs[4] / north_star."""
        super().__init__()
        self.dim = dim
        self.heads = heads
        self.separate_codebook_per_head = separate_codebook_per_head
        self.channel_last = channel_last

        head_dim = dim if codebook_dim is None else codebook_dim
        inner = head_dim * heads
        self.has_projections = inner != dim
        if self.has_projections:
            lin = nn.Linear(dim, inner)
            self.project_in = nn.Sequential(lin, nn.LayerNorm(inner)) if layernorm_after_project_in else lin
            self.project_out = nn.Linear(inner, dim)
        else:
            self.project_in = nn.Identity()
            self.project_out = nn.Identity()

        self.commitment_weight = commitment_weight
        self.has_commitment_loss = commitment_weight > 0.0
        self.commitment_use_cross_entropy_loss = commitment_use_cross_entropy_loss
        self.has_codebook_orthogonal_loss = orthogonal_reg_weight > 0.0
        self.orthogonal_reg_weight = orthogonal_reg_weight
        self.orthogonal_reg_active_codes_only = orthogonal_reg_active_codes_only
        self.orthogonal_reg_max_codes = orthogonal_reg_max_codes
        self.has_codebook_diversity_loss = codebook_diversity_loss_weight > 0.0
        self.codebook_diversity_loss_weight = codebook_diversity_loss_weight
        self.codebook_diversity_temperature = codebook_diversity_temperature

        assert not (codebook_params.ema_update and codebook_params.learnable_codebook), (
            "learnable codebook not compatible with EMA update"
        )
        assert 0 <= sync_update_v <= 1.0
        assert not (sync_update_v > 0.0 and not codebook_params.learnable_codebook), (
            "learnable codebook must be turned on"
        )
        self.sync_update_v = sync_update_v

        # ---- codebook sharded over the ranks of a process group (new capability: SURVEY 8e, BASELINE configs[4])
        self.shard_group = None
        self.shard_world, self.shard_rank = 1, 0
        self.codebook_size = codebook_params.codebook_size  # global K
        if codebook_shard_group is not None and codebook_shard_group is not False:
            assert dist.is_available() and dist.is_initialized(), "codebook_shard_group needs an initialised process group"
            assert codebook_shard_reduction in ("all_gather", "all_reduce") and codebook_shard_gather in ("owner", "replicated")
            self.shard_group = None if codebook_shard_group is True else codebook_shard_group
            self.shard_world = dist.get_world_size(self.shard_group)
            self.shard_rank = dist.get_rank(self.shard_group)
            unsupported = dict(commitment_use_cross_entropy_loss=commitment_use_cross_entropy_loss,
                               orthogonal_reg=orthogonal_reg_weight > 0.0, diversity_loss=codebook_diversity_loss_weight > 0.0,
                               in_place_codebook_optimizer=in_place_codebook_optimizer is not None,
                               learnable_codebook=codebook_params.learnable_codebook,
                               initialization_by_kmeans=codebook_params.initialization_by_kmeans,
                               stochastic_sampling=_stochastic_sampling_requested(codebook_params))
            bad = [k for k, v in unsupported.items() if v]
            if bad:
                raise NotImplementedError(f"a sharded codebook supports the search / quantize step / EMA update only, not {bad}")
            assert codebook_params.codebook_size % self.shard_world == 0, "codebook_size must divide evenly over the shard group"
            codebook_params = replace(codebook_params, codebook_size=codebook_params.codebook_size // self.shard_world)
            sync_codebook = False  # every rank owns different codes: nothing to average between ranks
        self.shard_reduction = codebook_shard_reduction
        self.shard_gather = codebook_shard_gather
        self._shard_table = None

        if sync_codebook is None:
            sync_codebook = _world_is_distributed()
        self.codebook_params = replace(
            codebook_params,
            dim=head_dim,
            num_codebooks=heads if separate_codebook_per_head else 1,
            learnable_codebook=self.has_codebook_orthogonal_loss or codebook_params.learnable_codebook,
            use_ddp=sync_codebook,
        )
        self.learnable_codebook = codebook_params.learnable_codebook
        self._codebook = Codebook(**asdict(self.codebook_params))
        # vector_quantize_pytorch.py:130-134: a factory that is handed the codebook's parameters
        self.in_place_codebook_optimizer = (
            in_place_codebook_optimizer(self._codebook.parameters()) if in_place_codebook_optimizer is not None else None
        )
        self.register_buffer("zero", torch.tensor(0.0), persistent=False)
        self._zero_loss = None

    # ------------------------------------------------------------------ codebook access (repaired w.r.t. the fork)
    @property
    def codebook(self):
        """[K, D] (or [heads, K, D] with per-head codebooks).  The fork reads a non-existent ``.embed``
        (vector_quantize_pytorch.py:142); this build exposes the real buffer."""
        codes = self._codebook.embeddings
        return codes if self.separate_codebook_per_head else codes[0]

    @codebook.setter
    def codebook(self, codes):
        if not self.separate_codebook_per_head:
            codes = codes[None]
        with torch.no_grad():
            self._codebook.embeddings.copy_(codes)

    def get_codes_from_indices(self, indices):
        codes = self.codebook
        if codes.ndim == 2:
            out = codes[indices]
        else:  # indices [b, ..., h] -> concatenate the per-head codes on the feature axis
            per_head = [codes[h][indices[..., h]] for h in range(codes.shape[0])]
            out = torch.cat(per_head, dim=-1)
        if not self.channel_last:
            out = out.movedim(-1, 1)
        return out

    def get_output_from_indices(self, indices):
        codes = self.get_codes_from_indices(indices)
        if not self.channel_last:
            return self.project_out(codes.movedim(1, -1)).movedim(-1, 1)
        return self.project_out(codes)

    # ------------------------------------------------------------------ codebook sharded over a process group
    def _reduce_keys(self, planes):
        """This rank's candidate planes [P, h, M] of packed keys -> the candidate planes of the whole codebook (identical on
        every rank): all G * P planes after the one-hop all-gather, or P planes reduced with MIN by the all-reduce.  The
        finalize takes the MIN over whatever planes it is handed."""
        planes = planes.contiguous()
        if self.shard_reduction == "all_gather":
            every = torch.empty((self.shard_world * planes.shape[0], *planes.shape[1:]), dtype=torch.int64, device=planes.device)
            dist.all_gather_into_tensor(every.view(-1), planes.view(-1), group=self.shard_group)
            return every
        dist.all_reduce(planes, op=dist.ReduceOp.MIN, group=self.shard_group)
        return planes

    def gather_table(self):
        """The full natural codebook [h, K, D], all-gathered from the shards and cached until this rank's shard changes
        (the ranks change their shards in lockstep: EMA steps and checkpoint loads are collective)."""
        cb = self._codebook
        key = cb.codes_state()
        if self._shard_table is None or self._shard_table[0] != key:
            local = cb.embeddings.detach().contiguous()  # [h, K/G, D]
            parts = torch.empty((self.shard_world * local.numel(),), dtype=local.dtype, device=local.device)
            dist.all_gather_into_tensor(parts, local.reshape(-1), group=self.shard_group)
            parts = parts.view(self.shard_world, *local.shape)
            self._shard_table = (key, parts.permute(1, 0, 2, 3).reshape(local.shape[0], -1, local.shape[-1]).contiguous())
        return self._shard_table[1]

    @torch.no_grad()
    def _sharded_search(self, flat, out_view, idx_view):
        """flat [h, M, D] (the same rows on every rank) -> (quantized rows [h, M, D], GLOBAL indices [h, M])."""
        from . import search

        cb, backend = self._codebook, search.get_backend()
        k_local = cb.codebook_size
        x = flat if flat.dtype == torch.float32 else flat.float()
        planes = backend.shard_keys(x, cb.embeddings.detach(), metric=cb.metric, idx_offset=self.shard_rank * k_local,
                                    packed=cb.packed_codes())
        planes = self._reduce_keys(planes if planes.dim() == 3 else planes[None])
        if self.shard_gather == "replicated":
            quant, idx = backend.finalize_keys(x, self.gather_table(), planes, metric=cb.metric)
        else:
            keys = planes[0] if planes.shape[0] == 1 else planes.amin(dim=0)
            idx = keys & 0xFFFFFFFF
            local = idx - self.shard_rank * k_local
            mine = (local >= 0) & (local < k_local)
            h = x.shape[0]
            rows = cb.embeddings.detach()[torch.arange(h, device=x.device)[:, None], local.clamp(0, k_local - 1)]
            quant = torch.where(mine[..., None], rows, torch.zeros((), dtype=rows.dtype, device=rows.device))
            dist.all_reduce(quant, group=self.shard_group)  # exactly one owner per row: x + 0 + ... is exact
        out_view.copy_(quant)
        idx_view.copy_(idx[..., None])
        return out_view, idx

    @torch.no_grad()
    def _sharded_ema_step(self, flat, idx):
        """EMA statistics of the codes THIS rank owns (rows whose winner lives here), then the shard's update."""
        from . import search

        cb = self._codebook
        k_local = cb.codebook_size
        local = idx - self.shard_rank * k_local
        mine = (local >= 0) & (local < k_local)
        hits, sums = search.get_backend().ema_accumulate(flat.float(), local.clamp(0, k_local - 1).contiguous(), k_local, mine)
        cb.ema_apply_shard(hits, sums, self.shard_group, self.codebook_size)
        cb.reseed_dead_codes(flat)

    # ------------------------------------------------------------------ forward
    def forward(self, x, indices=None, mask=None, freeze_codebook=False, return_loss_breakdown=False):
        return_loss = indices is not None
        orig_input = x
        single_vectors = x.ndim == 2
        if single_vectors:
            assert mask is None
            x = x[:, None, :]

        if not self.channel_last:
            x = x.movedim(1, -1)  # b d ... -> b ... d (view)
        spatial = tuple(x.shape[1:-1])
        batch = x.shape[0]
        n = 1
        for extent in spatial:
            n *= extent
        x = x.reshape(batch, n, x.shape[-1])  # b n d  (copy only if the permuted view cannot be flattened)
        x = self.project_in(x)

        heads, cb = self.heads, self._codebook
        head_dim = x.shape[-1] // heads
        x4 = x.reshape(batch, n, heads, head_dim)
        x4 = cb.transform_input(x4)
        wide_input = x4.dtype == torch.float64  # the reference's straight-through sum x + (q - x) promotes to the input's width
        plain_inference = not self.training and indices is None and mask is None and cb.is_initialized
        if x4.dtype != torch.float32 and not (plain_inference and x4.dtype in (torch.float16, torch.bfloat16)):
            x4 = x4.float()  # (2-byte rows of a plain inference forward are widened inside the search kernel instead)
        if not x4.is_contiguous():
            x4 = x4.contiguous()
        rows = batch * n
        # destination buffers in (row, head) order; the kernel writes through strided views of them
        q_buf = torch.empty((rows, heads, head_dim), dtype=torch.float32, device=x4.device)
        i_buf = torch.empty((rows, heads, 1), dtype=torch.int64, device=x4.device)
        if self.separate_codebook_per_head:
            flat = x4.view(rows, heads, head_dim).permute(1, 0, 2)  # [h, rows, d] strided view, no copy
            out_view, idx_view = q_buf.permute(1, 0, 2), i_buf.permute(1, 0, 2)
        else:
            flat = x4.view(1, rows * heads, head_dim)  # one codebook: heads are just more rows
            out_view, idx_view = q_buf.view(1, rows * heads, head_dim), i_buf.view(1, rows * heads, 1)

        def per_head_rows(t):
            """[rows, heads] (b, n, head order) -> the [H, M] arrangement of ``flat``'s rows."""
            return t.permute(1, 0) if self.separate_codebook_per_head else t.reshape(1, rows * heads)

        training = self.training
        use_ce = self.commitment_use_cross_entropy_loss
        want_loss = training and self.has_commitment_loss and not return_loss
        want_sq_err = want_loss and not use_ce
        flat_mask = None
        if mask is not None:
            per_row = mask.reshape(rows)
            flat_mask = per_head_rows(per_row[:, None].expand(rows, heads))

        def sampling_rows():
            """The rows as the reference orders them, for steps that pick rows by a random INDEX (k-means seeding, dead-code
            re-seeding): heads that share one codebook are flattened "(b h) n" there (vector_quantize_pytorch.py:217-219), while
            the search above runs on the (b n h) order of the buffer (the search itself is order-independent)."""
            if self.separate_codebook_per_head or heads == 1:
                return flat.detach()
            return x4.detach().view(batch, n, heads, head_dim).transpose(1, 2).reshape(1, rows * heads, head_dim)

        if not cb.is_initialized:
            seed_mask = flat_mask
            if flat_mask is not None and not (self.separate_codebook_per_head or heads == 1):
                seed_mask = mask.reshape(batch, 1, n).expand(batch, heads, n).reshape(1, rows * heads)
            cb.seed_with_kmeans(sampling_rows(), seed_mask)
            cb.is_initialized = True

        if training or return_loss:
            loss = torch.zeros(1, device=x.device, dtype=torch.float32)
        else:  # inference: nothing is ever added to it -- a persistent zero instead of a fill kernel per forward
            loss = _cached_zeros(self, "_zero_loss", (1,), x.device)
        commit_loss = diversity_loss = orthogonal_loss = inplace_loss = self.zero
        cb_grad_from_err = self.learnable_codebook and not freeze_codebook
        will_update = training and cb.ema_update and not freeze_codebook

        if self.in_place_codebook_optimizer is not None and training and not freeze_codebook:
            # vector_quantize_pytorch.py:234-259: one optimizer step on mse(quantize, x) with respect to the codebook,
            # then the codebook is searched again.  Only the search is needed from the first pass.
            with torch.no_grad():
                _, first_idx, _ = cb.quantize_flat(flat.detach())
            picked = cb.current_codes()[torch.arange(flat.shape[0], device=flat.device)[:, None], first_idx]
            err = (picked - flat.detach()) ** 2
            inplace_loss = err[flat_mask].mean() if flat_mask is not None else err.mean()
            inplace_loss.backward()
            self.in_place_codebook_optimizer.step()
            self.in_place_codebook_optimizer.zero_grad()
            if will_update:  # the reference's first Codebook.forward already ran its EMA step
                cb.ema_step(flat.detach(), first_idx, flat_mask, sample_pool=sampling_rows)
        # cross-entropy commitment: the search sweep also emits the row's log-sum-exp (no second sweep for the loss)
        ce_from_search = training and want_loss and use_ce
        ce_stats = None
        if self.shard_world > 1:
            assert mask is None and not return_loss, "a sharded codebook supports neither masks nor given indices"
            quant, idx = self._sharded_search(flat.detach(), out_view, idx_view)
            if training:
                out = flat + (quant - flat).detach()  # straight-through (vector_quantize_pytorch.py:273)
                if want_sq_err:
                    commit_loss = ((quant.detach() - flat) ** 2).mean()
                if will_update:
                    self._sharded_ema_step(flat.detach(), idx)
                    will_update = False
            else:
                out = quant
        elif mask is None:
            # the one native launch: search + gather + straight-through + squared error
            out, idx, sq_err, *rest = cb.quantize_flat(flat, ste=training, want_sq_err=want_sq_err,
                                                       codebook_grad_from_err=cb_grad_from_err, out=out_view,
                                                       idx=idx_view, want_lse=ce_from_search, frozen=freeze_codebook)
            ce_stats = rest[0] if rest else None
            if want_sq_err:
                commit_loss = (sq_err[0] / flat.numel()).to(torch.float32)
        else:
            out, idx, _, *rest = cb.quantize_flat(flat.detach() if not cb_grad_from_err else flat, ste=False, idx=idx_view,
                                                  want_lse=ce_from_search)
            ce_stats = rest[0] if rest else None
            if want_sq_err:
                target = out if cb_grad_from_err else out.detach()
                commit_loss = ((target - flat) ** 2)[flat_mask].mean()
            if training:
                out = flat + (out - flat).detach()
        if training and wide_input:
            out = out.double()
        if training and self.sync_update_v > 0.0:
            # eq. (21) of the vqtorch draft (vector_quantize_pytorch.py:275-279): same value, gradient scaled by 1 + v
            out = out + self.sync_update_v * (out - out.detach())

        # ---- consumers of the similarity matrix (rare): evaluated against the codebook the search
        #      used, i.e. BEFORE the EMA step below rewrites it
        needs_sims = return_loss or (training and ((want_loss and use_ce) or self.has_codebook_diversity_loss))
        if needs_sims:
            codes, live = cb.current_codes(), None
            if will_update:
                # the backward pass recomputes similarities from the pre-update codebook; the reference's autograd then
                # multiplies with the codebook as it is at backward time (losses.similarity_matrix: live_codes)
                codes, live = codes.detach().clone().requires_grad_(codes.requires_grad), cb.embeddings
            if return_loss:
                # cross entropy of the similarities against GIVEN codes (vector_quantize_pytorch.py:298-299)
                given = per_head_rows(indices.reshape(rows, heads).to(torch.int64))
                ce_given = losses.cross_entropy_to_codes(flat, codes, given, cb.metric, live)
            if training and want_loss and use_ce:
                target = idx_view[..., 0]
                if flat_mask is not None:
                    target = target.masked_fill(~flat_mask, -1)
                commit_loss = losses.cross_entropy_to_codes(flat, codes, target, cb.metric, live, stats=ce_stats)
            if training and self.has_codebook_diversity_loss and not return_loss:
                row_id = torch.arange(flat.shape[1], device=flat.device)
                position = (row_id % n) if (self.separate_codebook_per_head or heads == 1) else (row_id // heads) % n
                diversity_loss = losses.codebook_diversity_loss(flat, codes, cb.metric, self.codebook_diversity_temperature,
                                                                position, n, live)
                loss = loss + diversity_loss * self.codebook_diversity_loss_weight
        if want_loss:
            loss = loss + commit_loss * self.commitment_weight

        if will_update:
            cb.ema_step(flat.detach(), idx, flat_mask, sample_pool=sampling_rows)

        if return_loss:
            # the reference returns here, before heads are merged / projected back (vector_quantize_pytorch.py:298)
            if self.separate_codebook_per_head and heads > 1:
                quantize = out.reshape(heads, batch, n, head_dim)
            elif heads > 1:
                quantize = out.reshape(batch, n, heads, head_dim).permute(0, 2, 1, 3).reshape(1, batch * heads, n, head_dim)
            else:
                quantize = out.reshape(batch, n, head_dim)
            return quantize, ce_given

        ind_rows = i_buf.view(rows, heads)
        if training and want_loss and use_ce and mask is not None:
            ind_rows = ind_rows.masked_fill(~per_row[:, None], -1)  # the reference fills in place (vector_quantize_pytorch.py:344)

        if training and self.has_codebook_orthogonal_loss:
            # the fork reads a non-existent ``_codebook.embed`` here (vector_quantize_pytorch.py:367); repaired
            codes = cb.embeddings
            if self.orthogonal_reg_active_codes_only:
                assert not (heads > 1 and self.separate_codebook_per_head), (
                    "orthogonal regularization for only active codes not compatible with multi-headed with "
                    "separate codebooks yet"
                )
                used = torch.unique(ind_rows)
                codes = codes[:, used[used >= 0]]
            num_codes = codes.shape[-2]
            if self.orthogonal_reg_max_codes is not None and num_codes > self.orthogonal_reg_max_codes:
                pick = torch.randperm(num_codes, device=codes.device)[: self.orthogonal_reg_max_codes]
                codes = codes[:, pick]
            orthogonal_loss = losses.orthogonal_loss(codes)
            loss = loss + orthogonal_loss * self.orthogonal_reg_weight

        # ---- view the results back: rows are (b, n[, h]) ordered in both head modes
        if self.separate_codebook_per_head:
            quantize = out.permute(1, 0, 2).reshape(batch, n, heads * head_dim)  # a view when `out` is q_buf's
        else:
            quantize = out.reshape(batch, n, heads * head_dim)
        embed_ind = ind_rows.view(batch, n, heads)
        if heads == 1:
            embed_ind = embed_ind[..., 0]
            embed_ind = embed_ind.reshape(batch, *spatial)
        else:
            embed_ind = embed_ind.reshape(batch, *spatial, heads)
        if single_vectors:
            embed_ind = embed_ind[:, 0]

        quantize = self.project_out(quantize)
        quantize = quantize.reshape(batch, *spatial, quantize.shape[-1])
        if not self.channel_last:
            quantize = quantize.movedim(-1, 1)
        if single_vectors:
            quantize = quantize[:, 0]
        if mask is not None:
            quantize = torch.where(mask[..., None], quantize, orig_input)

        if not return_loss_breakdown:
            return quantize, embed_ind, loss
        return quantize, embed_ind, loss, LossBreakdown(commit_loss, diversity_loss, orthogonal_loss, inplace_loss)
