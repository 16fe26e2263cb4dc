"""``Codebook``: the stateful holder of the code vectors, re-written around ONE native search op.

Mirrors the constructor signature, buffer names (= checkpoint format: ``embeddings``, ``embed_avg``,
``cluster_size``) and return convention of the reference class
(/root/reference/vector_quantization/codebooks.py:81-435), but never materialises the ``[h, M, K]``
similarity / one-hot tensors the reference builds on every call (codebooks.py:386-390,
utils/general.py:129): search + gather (+ straight-through + squared error) is a single HIP launch.

What is native here: the forward search/gather (``search.quantize_rows``).  What is plain PyTorch on
the same device (training-state bookkeeping that comes AFTER the hot path): EMA statistics,
dead-code re-seeding, k-means seeding (which reuses the native search for its assignment step).
"""
from __future__ import annotations

from dataclasses import asdict, is_dataclass

import torch
import torch.distributed as dist
import torch.nn.functional as F
from torch import nn

from . import search
from .params import GumbelParams, KmeansParameters


def _identity(t):
    return t


def _unit_rows(t):
    return F.normalize(t, p=2, dim=-1)


_ROW_TRANSFORMS = {"identity": _identity, "l2norm": _unit_rows}


def _row_transform(name: str, what: str):
    if name not in _ROW_TRANSFORMS:
        # the reference does `raise "<str>"`, which surfaces as a TypeError (codebooks.py:110,117)
        raise TypeError(f"The option {name} for {what} is not implemented")
    return _ROW_TRANSFORMS[name]


def _default_codebook(h: int, k: int, d: int) -> torch.Tensor:
    """Same distribution as the reference's default (kaiming_uniform_ on [h, K, D]: fan_in = K * D)."""
    t = torch.empty(h, k, d)
    nn.init.kaiming_uniform_(t)
    return t


def _pick_rows(rows: torch.Tensor, count: int) -> torch.Tensor:
    """`count` rows of a [N, D] tensor: a random subset when N >= count, else draws with replacement."""
    n = rows.shape[0]
    if n >= count:
        sel = torch.randperm(n, device=rows.device)[:count]
    else:
        sel = torch.randint(0, n, (count,), device=rows.device)
    return rows[sel]


def _pick_rows_all_ranks(rows: torch.Tensor, count: int) -> torch.Tensor:
    """`count` rows drawn from the union of every rank's rows, IDENTICAL on all ranks (utils/distributed.py:56-78): rank 0
    splits `count` over the ranks in proportion to their row counts (sequential binomial draws), every rank samples its share
    locally, the shares are exchanged.  Replicas that seed or re-seed codes from this keep bit-identical codebooks."""
    world, rank = dist.get_world_size(), dist.get_rank()
    sizes = [torch.zeros((), dtype=torch.long, device=rows.device) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor(rows.shape[0], dtype=torch.long, device=rows.device))
    sizes = torch.stack(sizes)
    if rank == 0:
        probs = (sizes / sizes.sum()).cpu()
        left = probs.new_full((), count)
        remainder = probs.new_ones(())
        share = torch.empty_like(probs, dtype=torch.long)
        for i, p in enumerate(probs):
            drawn = torch.binomial(left, p / remainder)
            share[i] = drawn
            left -= drawn
            remainder -= p
        assert left == 0, f"invalid total count {left}"
        share = share.to(rows.device)
    else:
        share = torch.empty_like(sizes)
    dist.broadcast(share, src=0)
    share = share.tolist()
    mine = _pick_rows(rows, share[rank])
    parts = []
    for src, n in enumerate(share):
        part = mine if src == rank else rows.new_empty((n, *rows.shape[1:]))
        if n > 0:  # (every rank knows the shares: empty ones are skipped consistently)
            dist.broadcast(part, src=src)
        parts.append(part)
    return torch.cat(parts, dim=0)


class Codebook(nn.Module):
    def __init__(
        self,
        dim,
        codebook_size,
        num_codebooks=1,
        initialization_by_kmeans: bool = False,
        kmeans_params: KmeansParameters = None,
        decay: float = 0.8,
        eps_for_smoothing: float = 1e-5,
        threshold_ema_dead_code: int = 2,
        reset_cluster_size: int = None,
        use_ddp: bool = False,
        distributed_replace_codes: bool = True,
        learnable_codebook: bool = False,
        gumbel_params: GumbelParams = None,
        ema_update: bool = True,
        use_affine: bool = False,
        affine_params=None,
        transform_input: str = "identity",
        use_cosine_sim: bool = False,
        weights_regularization: str = "identity",
    ):
        super().__init__()
        self.transform_input = _row_transform(transform_input, "transform input function")
        self.weights_regularization = _row_transform(weights_regularization, "weights regularization")
        if use_affine:
            raise NotImplementedError("affine re-parameterisation is outside the MI355X hot-path build (SURVEY 2, #6)")

        self.dim = dim
        self.codebook_size = codebook_size
        self.num_codebooks = num_codebooks
        self.use_cosine_sim = use_cosine_sim
        self.metric = search.DOT if use_cosine_sim else search.EUCLID
        self.decay = decay
        self.ema_update = ema_update
        self.eps_for_smoothing = eps_for_smoothing
        self.threshold_ema_dead_code = threshold_ema_dead_code
        self.reset_cluster_size = threshold_ema_dead_code if reset_cluster_size is None else reset_cluster_size
        self.use_ddp = use_ddp
        self.distributed_replace_codes = distributed_replace_codes
        self.learnable_codebook = learnable_codebook
        self.use_affine = False

        if kmeans_params is None:
            kmeans_params = KmeansParameters() if initialization_by_kmeans else None
        self.kmeans_params = asdict(kmeans_params) if is_dataclass(kmeans_params) else kmeans_params
        if gumbel_params is None:
            gumbel_params = GumbelParams()
        self.gumbel_params = asdict(gumbel_params) if is_dataclass(gumbel_params) else dict(gumbel_params)

        assert not (use_ddp and num_codebooks > 1 and initialization_by_kmeans), (
            "kmeans init is not compatible with multiple codebooks in distributed environment for now"
        )

        start = torch.zeros(num_codebooks, codebook_size, dim) if initialization_by_kmeans else _default_codebook(
            num_codebooks, codebook_size, dim)
        start = self.weights_regularization(start)
        self.is_initialized = not initialization_by_kmeans
        self.register_buffer("cluster_size", torch.zeros(num_codebooks, codebook_size))
        self.register_buffer("embed_avg", start.clone())
        if learnable_codebook:
            self.embeddings = nn.Parameter(start)
        else:
            self.register_buffer("embeddings", start)
        # packed image of `embeddings` for the native search, rebuilt only when the codes change (see packed_codes)
        self._packed = None
        self._packed_key = None
        self._packed_epoch = 0

    # ------------------------------------------------------------------ helpers
    def _stochastic_requested(self) -> bool:
        """The reference always samples through ``sample_fn_training`` (codebooks.py:388), whose ``training`` flag is
        GumbelParams.training (default True) -- NOT the module's train / eval state."""
        g = self.gumbel_params
        return bool(g.get("training", True) and g.get("stochastic", False) and g.get("temperature", 1.0) > 0)

    def _sync_sum(self, t: torch.Tensor) -> torch.Tensor:
        if self.use_ddp and dist.is_available() and dist.is_initialized():
            dist.all_reduce(t)
        return t

    def current_codes(self) -> torch.Tensor:
        return self.embeddings if self.learnable_codebook else self.embeddings.detach()

    def codes_state(self):
        """Identity of the code values: storage, in-place version counter (``copy_`` / ``load_state_dict`` / optimizer steps
        bump it) and an epoch for writes the counter cannot see (``.data`` index writes, the native EMA kernel)."""
        e = self.embeddings
        if e.is_inference():  # built / loaded under torch.inference_mode(): no version counter -> never reuse a packed image
            return (e.data_ptr(), object(), self._packed_epoch, e.device, tuple(e.shape))
        return (e.data_ptr(), e._version, self._packed_epoch, e.device, tuple(e.shape))

    def invalidate_packed(self):
        """Call after changing ``embeddings`` through ``.data`` or a raw pointer (every writer in this package does)."""
        self._packed_epoch += 1

    def packed_codes(self):
        """Packed image [h, packed_floats] of the codes for the native search (None for backends without one), cached:
        an inference forward does not re-pack an unchanged codebook (13 us at K=1024 D=256, 121 us at K=65536 D=512)."""
        backend = search.get_backend()
        if not getattr(backend, "uses_packed", False):
            return None
        if torch.compiler.is_compiling():  # traced: the pack is a node of the graph (no identity-keyed cache inside a trace)
            return backend.pack(self.embeddings.detach().contiguous(), self.metric)
        if not self.embeddings.is_cuda:
            return None
        key = (self.codes_state(), self.metric)
        if self._packed_key != key:
            with torch.no_grad():
                self._packed = backend.pack(self.embeddings.detach().contiguous(), self.metric)
            self._packed_key = key
        return self._packed

    # ------------------------------------------------------------------ the hot path
    def quantize_flat(self, flat: torch.Tensor, *, ste: bool = False, want_sq_err: bool = False,
                      codebook_grad_from_err: bool = False, out=None, idx=None, want_lse: bool = False,
                      frozen: bool = False):
        """flat [h, M, D] (strided rows fine) -> (out [h, M, D], idx [h, M] int64, sq_err [1] float64 | None).
        ``want_lse``: a fourth element (lse [h, M], target_logit [h, M]) -- log-sum-exp of the row's similarities and the
        similarity of the chosen code, from the same sweep -- or None when the sampling is stochastic.
        ``frozen``: the caller will NOT run the EMA update after this search (freeze_codebook)."""
        if self._stochastic_requested():
            res = self._quantize_stochastic(flat, ste=ste, want_sq_err=want_sq_err,
                                            codebook_grad_from_err=codebook_grad_from_err, idx=idx)
            return (*res, None) if want_lse else res
        codes = self.current_codes()
        packed = self.packed_codes()
        if (torch.is_grad_enabled() and flat.requires_grad and not codes.requires_grad and self.training and self.ema_update
                and not frozen):
            # the EMA step that follows rewrites the codebook in place; the backward pass (commitment loss: 2 (x - c))
            # must see the codes this forward used, as the reference's autograd graph does (it holds `quantize` by value)
            codes = codes.clone()
        # (the search sweep emits the log-sum-exp only for rows one launch holds; wider rows leave it to the loss)
        lse_here = want_lse and flat.shape[-1] <= search.LSE_MAX_DIM
        res = search.quantize_rows(flat, codes[:, None], metric=self.metric, ste=ste, want_sq_err=want_sq_err,
                                   codebook_grad_from_err=codebook_grad_from_err, out=out, idx=idx, want_lse=lse_here,
                                   packed=packed)
        out, idx, sq_err = res[:3]
        if want_lse and not lse_here:
            return out, idx[..., 0], sq_err, None
        if want_lse:
            best = res[3]["best"][..., 0]
            return out, idx[..., 0], sq_err, (res[3]["lse"], best if self.use_cosine_sim else -best)
        return out, idx[..., 0], sq_err

    def _quantize_stochastic(self, flat, *, ste, want_sq_err, codebook_grad_from_err, idx=None):
        """Gumbel-max sampling of the code (utils/general.py:106-129): ind = argmax(similarities / temperature + g),
        g = -log(-log(u)).  RNG-dependent, so no parity with the reference's draws is possible; the similarities come
        from the native kernel in bounded row chunks and the noise from torch's generator on the tensor's device.
        Straight-through / reinmax relaxations (gradients through the softmax) are not provided."""
        from . import losses

        g = self.gumbel_params
        if g.get("straight_through", False) or g.get("reinmax", False):
            raise NotImplementedError("straight-through / reinmax Gumbel relaxations are outside the MI355X hot-path "
                                      "build (SURVEY 2, #4); plain stochastic sampling is supported")
        codes = self.current_codes()
        h, m, _ = flat.shape
        x = flat.float()
        step = losses._rows_per_chunk(h, self.codebook_size)
        ind = torch.empty((h, m), dtype=torch.int64, device=flat.device)
        eps = 1e-5  # the reference's log(t) clamps at 1e-5 (utils/general.py:25-26)
        with torch.no_grad():
            for r0 in range(0, m, step):
                sims = losses.similarity_matrix(x[:, r0:r0 + step].detach(), codes.detach(), self.metric)
                noise = torch.zeros_like(sims).uniform_(0, 1)
                gumbel = -(-noise.clamp(min=eps).log()).clamp(min=eps).log()
                ind[:, r0:r0 + step] = (sims / g.get("temperature", 1.0) + gumbel).argmax(dim=-1)
        if idx is not None:
            idx.copy_(ind[..., None])
        picked = codes[torch.arange(h, device=flat.device)[:, None], ind]  # [h, m, d]; differentiable w.r.t. a learnable codebook
        sq_err = None
        if want_sq_err:
            target = picked if codebook_grad_from_err else picked.detach()
            sq_err = ((target - x) ** 2).sum(dtype=torch.float64).reshape(1)
        out = x + (picked - x).detach() if ste else picked
        return out, ind, sq_err

    def similarities(self, flat: torch.Tensor) -> torch.Tensor:
        """The [h, M, K] matrix the reference returns on every call (codebooks.py:386,435), on demand
        (vq_similarities_f32: the very values the search compares).  The losses that consume it never ask for the
        whole matrix -- see losses.py."""
        from . import losses

        return losses.similarity_matrix(flat, self.current_codes(), self.metric)

    def forward(self, x, mask=None, freeze_codebook=False, return_similarities=True):
        """(quantize, embed_ind, similarities) like the reference (codebooks.py:351,435).  A direct caller gets the full
        ``[h, ..., K]`` similarity tensor, as the reference returns it; pass ``return_similarities=False`` to skip its
        materialisation (the modules of this package never call this method: they use ``quantize_flat``)."""
        squeeze_head = x.ndim < 4
        x = x.float()
        if squeeze_head:
            x = x.unsqueeze(0)
        h, d = x.shape[0], x.shape[-1]
        lead = x.shape[1:-1]
        flat = x.reshape(h, -1, d)
        flat_mask = None
        if mask is not None:
            reps = flat.shape[1] // (mask.shape[0] * mask.shape[1])
            flat_mask = mask[:, None, :].expand(mask.shape[0], reps, mask.shape[1]).reshape(1, -1).expand(h, -1)

        if not self.is_initialized:
            self.seed_with_kmeans(flat, flat_mask)
            self.is_initialized = True

        out, idx, _ = self.quantize_flat(flat, frozen=freeze_codebook)
        sims = self.similarities(flat).reshape(h, *lead, self.codebook_size) if return_similarities else None

        if self.training and self.ema_update and not freeze_codebook:
            self.ema_step(flat.detach(), idx, flat_mask)

        quantize = out.reshape(h, *lead, d)
        embed_ind = idx.reshape(h, *lead)
        if squeeze_head:
            quantize, embed_ind = quantize[0], embed_ind[0]  # `sims` keeps the head dim, like the reference (codebooks.py:433)
        return quantize, embed_ind, sims

    # ------------------------------------------------------------------ training-state bookkeeping
    @torch.no_grad()
    def ema_step(self, flat: torch.Tensor, idx: torch.Tensor, flat_mask=None, sample_pool=None):
        """Exponential-moving-average codebook update + dead-code re-seeding (codebooks.py:399-426),
        expressed with index arithmetic instead of the reference's [h, M, K] one-hot products.
        ``sample_pool``: the rows re-seeding draws from, in the REFERENCE's row order (a tensor or a callable returning it;
        default ``flat``) -- the draw is an index into them, so the order matters (shared codebook with several heads)."""
        hits, sums = search.get_backend().ema_accumulate(flat, idx, self.codebook_size, flat_mask)
        self.ema_apply(hits, sums)
        self.reseed_dead_codes(sample_pool if sample_pool is not None else flat)

    @torch.no_grad()
    def ema_apply(self, hits: torch.Tensor, sums: torch.Tensor):
        """Second half of the EMA step from ready statistics (hits [h, K], sums [h, K, D]): replica sync, lerp, Laplace
        smoothing, normalise -- codebooks.py:410-425."""
        hits, sums = hits.contiguous(), sums.contiguous()
        self._sync_sum(hits)
        self._sync_sum(sums)
        search.get_backend().ema_update(self.cluster_size.data, self.embed_avg.data, self.embeddings.data, hits, sums,
                                        decay=self.decay, eps=self.eps_for_smoothing,
                                        l2norm=self.weights_regularization is _unit_rows)
        self.invalidate_packed()

    @torch.no_grad()
    def ema_apply_shard(self, hits: torch.Tensor, sums: torch.Tensor, group, k_total: int):
        """EMA step of ONE SHARD of a codebook that is sharded over ``group`` (codebooks.py:410-425 for the codes this rank
        owns).  The statistics of a code live on its owner only, so they need no reduction; the Laplace smoothing
        normalises with the cluster sizes of the WHOLE codebook, so the per-shard totals are all-reduced."""
        w = 1.0 - self.decay
        self.cluster_size.data.lerp_(hits, w)
        self.embed_avg.data.lerp_(sums, w)
        total = self.cluster_size.data.sum(dim=-1, keepdim=True)
        dist.all_reduce(total, group=group)
        smoothed = (self.cluster_size.data + self.eps_for_smoothing) / (total + k_total * self.eps_for_smoothing) * total
        fresh = self.weights_regularization(self.embed_avg.data / smoothed[..., None])
        self.embeddings.data.copy_(fresh)
        self.invalidate_packed()

    @torch.no_grad()
    def reseed_dead_codes(self, flat):
        """``flat`` [h, M, D], or a callable producing it (evaluated only if some code actually expired)."""
        if self.threshold_ema_dead_code == 0:
            return
        dead = self.cluster_size < self.threshold_ema_dead_code
        if not bool(dead.any()):
            return
        if callable(flat):
            flat = flat()
        pool = self.weights_regularization(flat)
        for head in range(pool.shape[0]):
            n_dead = int(dead[head].sum().item())
            if n_dead == 0:
                continue
            kmeans_sync = (self.kmeans_params or {}).get("sync", True)
            spread = self.use_ddp and dist.is_available() and dist.is_initialized()
            if spread and kmeans_sync and self.distributed_replace_codes:
                picked = _pick_rows_all_ranks(pool[head], n_dead)  # the same replacement vectors on every replica
            else:
                picked = _pick_rows(pool[head], n_dead)
                if spread and not self.distributed_replace_codes:  # codebooks.py:236-237: average the replicas' draws
                    dist.all_reduce(picked)
                    picked = picked / dist.get_world_size()
            self.embeddings.data[head][dead[head]] = picked
            self.invalidate_packed()
            self.cluster_size.data[head][dead[head]] = self.reset_cluster_size
            self.embed_avg.data[head][dead[head]] = picked * self.reset_cluster_size

    @torch.no_grad()
    def seed_with_kmeans(self, flat: torch.Tensor, flat_mask=None):
        """Lloyd iterations on the first batch (utils/kmeans.py:38-120); the assignment step is the
        native search kernel."""
        if flat_mask is not None:
            h = flat.shape[0]
            flat = flat[flat_mask].reshape(h, -1, flat.shape[-1])
        h, m, d = flat.shape
        k = self.codebook_size
        iters = (self.kmeans_params or {}).get("iter", 10)
        sync = self.use_ddp and (self.kmeans_params or {}).get("sync", True)
        # utils/kmeans.py:82-118: the first centroids are RAW sampled rows and the per-cluster means are means of the RAW rows;
        # with the cosine similarity only the new centroids are L2-normalised (the rows are not, unless the module's
        # transform_input already did it)
        data = flat
        pick = _pick_rows_all_ranks if (sync and dist.is_available() and dist.is_initialized()) else _pick_rows
        means = torch.stack([pick(data[i], k) for i in range(h)])
        counts = torch.zeros((h, k), dtype=flat.dtype, device=flat.device)
        for _ in range(iters):
            idx, _best, _ = search.nearest_with_distance(data, means, metric=self.metric)
            counts, sums = search.get_backend().ema_accumulate(data.contiguous(), idx.contiguous(), k)
            if sync and dist.is_initialized():
                dist.all_reduce(counts)
                dist.all_reduce(sums)
            fresh = sums / counts.clamp(min=1.0)[..., None]
            if self.use_cosine_sim:
                fresh = _unit_rows(fresh)
            means = torch.where((counts == 0)[..., None], means, fresh)
        self.embeddings.data.copy_(means)
        self.invalidate_packed()
        self.embed_avg.data.copy_(means * counts[..., None])
        self.cluster_size.data.copy_(counts)
