"""Codebook-sharded nearest search over one node's GPUs (new capability; not in the reference).

Rank g of the process group holds rows ``[g*K/G, (g+1)*K/G)`` of a codebook that is searched as a whole:

    1. every rank searches ITS shard for all rows (native kernel) and emits one packed signed 64-bit key per
       row: (order image of the winning value) << 32 | (global code index);
    2. ONE collective, two interchangeable forms (``reduction=``):
       ``"all_reduce"``  ``all_reduce(keys, op=MIN)`` (RCCL ``ncclMin`` on int64; 8 bytes per row; a ring on xGMI, i.e.
                         2 (G - 1) latency-bound steps over one ~153 GB/s link each);
       ``"all_gather"``  every rank sends its M keys to every other rank in ONE hop over all 7 xGMI links
                         (``all_gather_into_tensor``, 8 (G - 1) M bytes received per rank) and takes the min over the G
                         candidates locally -- the better shape while M * 8 B is latency-bound.
       Equal distances resolve to the lowest GLOBAL index -- the reference's first-argmax semantics
       (utils/general.py:128) -- because the index sits in the low word, in either form;
    3. finalize: decode, gather ``codebook[idx]``.  The gather table is either a replicated full copy
       (``gather="replicated"``: 128 MiB at K=65536, D=512 is nothing in 288 GB) or stays sharded
       (``gather="owner"``: the owner rank contributes the row, everybody else zeros, and a SUM all-reduce
       delivers it: x + 0 is exact).

Parity target: the single-process reference with the full codebook.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.distributed as dist

from . import native

KEY_INIT = 0x7FFFFFFFFFFFFFFF


class _NativeShardOps:
    """The device steps, through the C ABI."""

    @staticmethod
    def prepare(shard, metric):
        """Pack the (static) shard once: the codebook of a ShardedCodebookSearch does not change between calls."""
        return native.pack_codebooks(shard[None].contiguous(), metric)

    @staticmethod
    def local_keys(x, shard, metric, idx_offset, packed=None):
        """-> candidate planes [P, M]: one launch, no init, no atomics (vq_search_key_planes_f32)."""
        return native.search_key_planes(x[None], shard[None], metric=metric, idx_offset=idx_offset, packed=packed)[:, 0]

    @staticmethod
    def finalize(x, table, keys, metric, ste, want_sq_err, out=None, idx=None, best=None):
        """keys: candidate planes [C, M] (MIN taken inside the finalize kernel)."""
        r = native.finalize_keys(x[None], table[None], keys[:, None], metric=metric, ste=ste, want_sq_err=want_sq_err,
                                 out=None if out is None else out[None], idx=None if idx is None else idx[None],
                                 best=None if best is None else best[None])
        return r["out"][0], r["idx"][0], r["best"][0], (r["sq_err"] if want_sq_err else None)


class _Pending:
    """An exchange in flight: ``wait()`` orders the current stream behind it (RCCL) and returns the candidate planes."""

    def __init__(self, work, planes):
        self.work, self.planes = work, planes

    def wait(self):
        if self.work is not None:
            self.work.wait()
        return self.planes


class ShardedCodebookSearch:
    """Nearest-code search against a codebook sharded over the ranks of ``group``.

    ``shard`` is this rank's [K/G, D] slice (rank order = index order).  ``full_codebook`` (optional) is a
    replicated [K, D] copy used only for the final gather.

    One step = ONE search launch (K split over the workgroups that fill the chip, every split storing into its own key
    plane: no init launch, no atomics) -> ONE collective on the planes -> ONE finalize launch that takes the MIN over the
    candidate planes while it gathers.  ``overlap_rows``: batches of at least that many rows are cut in two halves so that
    the exchange of the first half runs under the search of the second (the collective is asynchronous on RCCL's stream).
    """

    def __init__(self, shard: torch.Tensor, *, use_cosine_sim: bool = False, group=None,
                 full_codebook: Optional[torch.Tensor] = None, ops=None, reduction: str = "all_reduce",
                 overlap_rows: Optional[int] = 32768):
        assert reduction in ("all_reduce", "all_gather"), reduction
        self.reduction = reduction
        self.group = group
        self.metric = native.DOT if use_cosine_sim else native.EUCLID
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if self.world > 1 else 0
        self.shard = shard.contiguous().float()
        self.k_local = shard.shape[0]
        self.k_total = self.k_local * self.world
        self.full = full_codebook.contiguous().float() if full_codebook is not None else None
        self.ops = ops if ops is not None else _NativeShardOps
        self.packed = self.ops.prepare(self.shard, self.metric) if hasattr(self.ops, "prepare") else None
        self.overlap_rows = overlap_rows

    # ------------------------------------------------------------------ the three steps
    def local_planes(self, x: torch.Tensor) -> torch.Tensor:
        """x [M, D] fp32 -> this shard's candidate planes [P, M] int64."""
        if self.packed is not None:
            keys = self.ops.local_keys(x, self.shard, self.metric, self.rank * self.k_local, self.packed)
        else:
            keys = self.ops.local_keys(x, self.shard, self.metric, self.rank * self.k_local)
        return keys if keys.dim() == 2 else keys[None]

    def exchange(self, planes: torch.Tensor, async_op: bool = False) -> _Pending:
        """planes [P, M] of this rank -> candidate planes of the whole codebook (identical on every rank): all G * P planes
        after the one-hop all-gather, or P planes reduced with MIN by the all-reduce."""
        if self.world == 1:
            return _Pending(None, planes)
        planes = planes.contiguous()
        if self.reduction == "all_gather":
            every = torch.empty((self.world * planes.shape[0], planes.shape[1]), dtype=torch.int64, device=planes.device)
            work = dist.all_gather_into_tensor(every.view(-1), planes.view(-1), group=self.group, async_op=async_op)
            return _Pending(work if async_op else None, every)
        work = dist.all_reduce(planes, op=dist.ReduceOp.MIN, group=self.group, async_op=async_op)
        return _Pending(work if async_op else None, planes)

    def reduce_keys(self, keys: torch.Tensor) -> torch.Tensor:
        """keys [M] or planes [P, M] of this rank's shard -> ONE plane [M]: the element-wise minimum over planes and ranks."""
        planes = self.exchange(keys if keys.dim() == 2 else keys[None]).wait()
        return planes[0] if planes.shape[0] == 1 else planes.amin(dim=0)

    def search_keys(self, x: torch.Tensor) -> torch.Tensor:
        """x [M, D] (identical on every rank) -> reduced keys [M] int64 (identical on every rank)."""
        return self.reduce_keys(self.local_planes(x.float()))

    def quantize_local_rows(self, x_local: torch.Tensor, *, ste: bool = False, want_sq_err: bool = False):
        """Data-parallel callers: every rank holds ITS OWN rows x_local [M_local, D] (same M_local on all ranks) and
        wants them quantized against the whole sharded codebook ("tokens start rank-local").

        all-gather the rows (M_local * D * 4 B per rank) -> every rank searches its shard for ALL rows ->
        reduce-scatter(MIN) of the packed keys, so each rank receives exactly the reduced keys of its own rows
        (8 B per row on the wire) -> local finalize.  Needs the replicated gather table (``full_codebook``).
        -> (quantized [M_local, D], idx [M_local] GLOBAL, best [M_local], sq_err | None) for this rank's rows."""
        x_local = x_local.float().contiguous()
        if self.world == 1:
            return self(x_local, ste=ste, want_sq_err=want_sq_err)
        assert self.full is not None, "quantize_local_rows needs full_codebook for the local gather"
        m_local, d = x_local.shape
        rows = torch.empty((self.world * m_local, d), dtype=torch.float32, device=x_local.device)
        dist.all_gather_into_tensor(rows, x_local, group=self.group)
        planes = self.local_planes(rows)
        keys = planes[0] if planes.shape[0] == 1 else planes.amin(dim=0)
        mine = torch.empty((m_local,), dtype=torch.int64, device=x_local.device)
        try:
            dist.reduce_scatter_tensor(mine, keys.contiguous(), op=dist.ReduceOp.MIN, group=self.group)
        except (RuntimeError, NotImplementedError):  # backends without reduce-scatter (gloo): all-reduce, keep our slice
            dist.all_reduce(keys, op=dist.ReduceOp.MIN, group=self.group)
            mine = keys[self.rank * m_local:(self.rank + 1) * m_local].contiguous()
        return self.ops.finalize(x_local, self.full, mine[None], self.metric, ste, want_sq_err)

    def _owner_gather(self, x, planes, ste, want_sq_err):
        """The table stays sharded: the owner of a row's winner contributes the code, everybody else zeros, and a SUM
        all-reduce delivers it (x + 0 is exact).  Pure tensor ops, no host synchronisation."""
        keys = planes[0] if planes.shape[0] == 1 else planes.amin(dim=0)
        idx = keys & 0xFFFFFFFF
        local = idx - self.rank * self.k_local
        mine = (local >= 0) & (local < self.k_local)
        rows = self.shard[local.clamp(0, self.k_local - 1)]
        q = torch.where(mine[:, None], rows, torch.zeros((), dtype=rows.dtype, device=rows.device))
        dist.all_reduce(q, op=dist.ReduceOp.SUM, group=self.group)
        sq_err = ((q - x).double() ** 2).sum().reshape(1) if want_sq_err else None
        out = x + (q - x) if ste else q
        return out, idx, None, sq_err

    def __call__(self, x: torch.Tensor, *, ste: bool = False, want_sq_err: bool = False):
        """-> (quantized [M, D], idx [M] int64 GLOBAL indices, best [M], sq_err | None)."""
        x = x.float()
        replicated = self.full is not None or self.world == 1
        table = (self.full if self.full is not None else self.shard) if replicated else None
        m = x.shape[0]
        cut = (m // 2 + 255) // 256 * 256
        if (replicated and self.world > 1 and self.overlap_rows is not None and m >= self.overlap_rows and 0 < cut < m
                and not want_sq_err):
            # two halves: the exchange of the first half travels while the second half is being searched
            out = torch.empty((m, x.shape[1]), dtype=torch.float32, device=x.device)
            idx = torch.empty((m,), dtype=torch.int64, device=x.device)
            best = torch.empty((m,), dtype=torch.float32, device=x.device)
            pend = []
            for a, b in ((0, cut), (cut, m)):
                pend.append((a, b, self.exchange(self.local_planes(x[a:b]), async_op=True)))
            for a, b, pnd in pend:
                self.ops.finalize(x[a:b], table, pnd.wait(), self.metric, ste, False, out=out[a:b], idx=idx[a:b], best=best[a:b])
            return out, idx, best, None
        planes = self.exchange(self.local_planes(x)).wait()
        if replicated:
            return self.ops.finalize(x, table, planes, self.metric, ste, want_sq_err)
        return self._owner_gather(x, planes, ste, want_sq_err)

    def profile_phases(self, x: torch.Tensor, steps: int = 10):
        """Per-phase device time of one step (each phase looped on its own between synchronisations) and the host time it
        takes to ENQUEUE a whole step -- what explains a measured scaling factor: -> dict(search_ms, exchange_ms,
        finalize_ms, host_us, step_ms)."""
        import time

        x = x.float()
        dev = x.device
        sync = (lambda: torch.cuda.synchronize(dev)) if x.is_cuda else (lambda: None)
        table = self.full if self.full is not None else self.shard

        def timed(fn):
            fn()
            sync()
            if self.world > 1:
                dist.barrier(group=self.group)
            t0 = time.perf_counter()
            for _ in range(steps):
                r = fn()
            sync()
            return (time.perf_counter() - t0) / steps * 1e3, r

        search_ms, planes = timed(lambda: self.local_planes(x))
        exchange_ms, cand = timed(lambda: self.exchange(planes.clone()).wait())
        exchange_ms -= timed(lambda: planes.clone())[0]
        if self.full is not None or self.world == 1:
            finalize_ms, _ = timed(lambda: self.ops.finalize(x, table, cand, self.metric, False, False))
        else:
            finalize_ms, _ = timed(lambda: self._owner_gather(x, cand, False, False))
        sync()
        if self.world > 1:
            dist.barrier(group=self.group)
        t0 = time.perf_counter()
        for _ in range(steps):
            self(x)
        host_us = (time.perf_counter() - t0) / steps * 1e6  # enqueue only: nothing waited for yet
        sync()
        step_ms = (time.perf_counter() - t0) / steps * 1e3
        return dict(search_ms=round(search_ms, 4), exchange_ms=round(max(exchange_ms, 0.0), 4), finalize_ms=round(finalize_ms, 4),
                    host_us=round(host_us, 1), step_ms=round(step_ms, 4), key_planes=int(planes.shape[0]))
