"""``torch.library`` registration of the native op ("registered to PyTorch as ``torch.ops.<ns>.vq_nearest``").

The transport stays the C ABI over ctypes (``native.py``); these are the *same* calls given a dispatcher identity and a
fake (meta) implementation, so that ``torch.compile`` / ``torch.export`` can trace a module forward THROUGH the search
instead of breaking the graph at an opaque Python call:

    torch.ops.vq_mi355x.pack(cb, metric) -> packed images
    torch.ops.vq_mi355x.quantize_into(x, cb, packed, out, idx, metric, ste, want_sq_err, share, per_head) -> sq_err
        (writes the quantized rows and the indices into the caller's -- possibly strided -- ``out`` / ``idx`` views)

Eager forwards keep calling ``native.quantize`` directly (a custom-op dispatch costs tens of microseconds of host time,
which is most of a small launch); the modules switch to these ops only while being compiled
(``torch.compiler.is_compiling()``).  There is still no CPU implementation: the real kernels raise on CPU tensors.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import native

_LIB_NS = "vq_mi355x"


@torch.library.custom_op(f"{_LIB_NS}::pack", mutates_args=())
def pack(cb: torch.Tensor, metric: int) -> torch.Tensor:
    return native.pack_codebooks(cb.contiguous(), metric)


@pack.register_fake
def _(cb, metric):
    k, d = cb.shape[-2], cb.shape[-1]
    n = cb.numel() // max(1, k * d)
    return cb.new_empty((n, native.packed_floats(int(k), int(d))))


@torch.library.custom_op(f"{_LIB_NS}::quantize_into", mutates_args=("out", "idx"))
def quantize_into(x: torch.Tensor, cb: torch.Tensor, packed: Optional[torch.Tensor], out: torch.Tensor, idx: torch.Tensor,
                  metric: int, ste: bool, want_sq_err: bool, share: bool, per_head: bool) -> torch.Tensor:
    """x [H, M, D], cb [H, Q|1, K, D], out [H, M, D] / idx [H, M, Q] destination views -> sq_err ([Q] or [H, Q] float64;
    zeros when not requested)."""
    r = native.quantize(x, cb, metric=metric, ste=ste, want_sq_err=want_sq_err, want_best=False, packed=packed,
                        stages_share_codebook=share, out=out, idx=idx, sq_err_per_head=per_head)
    if r["sq_err"] is not None:
        return r["sq_err"]
    q = idx.shape[-1]
    return torch.zeros((x.shape[0], q) if per_head else (q,), dtype=torch.float64, device=x.device)


@quantize_into.register_fake
def _(x, cb, packed, out, idx, metric, ste, want_sq_err, share, per_head):
    q = idx.shape[-1]
    return x.new_empty((x.shape[0], q) if per_head else (q,), dtype=torch.float64)
