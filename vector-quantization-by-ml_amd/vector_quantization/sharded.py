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
    """The three device steps, through the C ABI."""

    @staticmethod
    def prepare(shard, metric):
        """Pack the (static) shard once: the codebook of a ShardedCodebookSearch does not change between calls."""
        return native.pack_codebooks(shard[None].contiguous(), metric)

    @staticmethod
    def local_keys(x, shard, metric, idx_offset, packed=None):
        keys = torch.empty((1, x.shape[0]), dtype=torch.int64, device=x.device)
        native.keys_init(keys)
        native.search_keys(x[None], shard[None], keys, metric=metric, idx_offset=idx_offset, packed=packed)
        return keys[0]

    @staticmethod
    def finalize(x, table, keys, metric, ste, want_sq_err):
        r = native.finalize_keys(x[None], table[None], keys[None], metric=metric, ste=ste, want_sq_err=want_sq_err)
        return r["out"][0], r["idx"][0], r["best"][0], (r["sq_err"] if want_sq_err else None)

    @staticmethod
    def decode(keys, metric):
        idx = keys & 0xFFFFFFFF
        return idx


class ShardedCodebookSearch:
    """Nearest-code search against a codebook sharded over the ranks of ``group``.

    ``shard`` is this rank's [K/G, D] slice (rank order = index order).  ``full_codebook`` (optional) is a
    replicated [K, D] copy used only for the final gather.
    """

    def __init__(self, shard: torch.Tensor, *, use_cosine_sim: bool = False, group=None,
                 full_codebook: Optional[torch.Tensor] = None, ops=None, reduction: str = "all_reduce"):
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

    def search_keys(self, x: torch.Tensor) -> torch.Tensor:
        """x [M, D] (identical on every rank) -> reduced keys [M] int64 (identical on every rank)."""
        if self.packed is not None:
            keys = self.ops.local_keys(x.float(), self.shard, self.metric, self.rank * self.k_local, self.packed)
        else:
            keys = self.ops.local_keys(x.float(), self.shard, self.metric, self.rank * self.k_local)
        return self.reduce_keys(keys)

    def reduce_keys(self, keys: torch.Tensor) -> torch.Tensor:
        """keys [M] int64 of this rank's shard -> the element-wise minimum over the ranks (identical everywhere)."""
        if self.world == 1:
            return keys
        if self.reduction == "all_gather":
            every = torch.empty((self.world * keys.shape[0],), dtype=torch.int64, device=keys.device)
            dist.all_gather_into_tensor(every, keys.contiguous(), group=self.group)
            return every.view(self.world, keys.shape[0]).amin(dim=0)
        dist.all_reduce(keys, op=dist.ReduceOp.MIN, group=self.group)
        return keys

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
        if self.packed is not None:
            keys = self.ops.local_keys(rows, self.shard, self.metric, self.rank * self.k_local, self.packed)
        else:
            keys = self.ops.local_keys(rows, self.shard, self.metric, self.rank * self.k_local)
        mine = torch.empty((m_local,), dtype=torch.int64, device=x_local.device)
        try:
            dist.reduce_scatter_tensor(mine, keys, op=dist.ReduceOp.MIN, group=self.group)
        except (RuntimeError, NotImplementedError):  # backends without reduce-scatter (gloo): all-reduce, keep our slice
            dist.all_reduce(keys, op=dist.ReduceOp.MIN, group=self.group)
            mine = keys[self.rank * m_local:(self.rank + 1) * m_local].contiguous()
        return self.ops.finalize(x_local, self.full, mine, self.metric, ste, want_sq_err)

    def __call__(self, x: torch.Tensor, *, ste: bool = False, want_sq_err: bool = False):
        """-> (quantized [M, D], idx [M] int64 GLOBAL indices, best [M], sq_err | None)."""
        x = x.float()
        keys = self.search_keys(x)
        if self.full is not None or self.world == 1:
            table = self.full if self.full is not None else self.shard
            return self.ops.finalize(x, table, keys, self.metric, ste, want_sq_err)
        # owner-contributes gather: rows whose winner lives elsewhere are zero here
        idx = keys & 0xFFFFFFFF
        local = idx - self.rank * self.k_local
        mine = (local >= 0) & (local < self.k_local)
        q = torch.zeros((x.shape[0], self.shard.shape[1]), dtype=torch.float32, device=x.device)
        q[mine] = self.shard[local[mine]]
        dist.all_reduce(q, op=dist.ReduceOp.SUM, group=self.group)
        sq_err = ((q - x).double() ** 2).sum().reshape(1) if want_sq_err else None
        out = x + (q - x) if ste else q
        return out, idx, None, sq_err
