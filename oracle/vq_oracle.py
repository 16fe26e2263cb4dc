"""CPU ORACLE (numpy + the C restatement in vq_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module.  Nothing under ``vector-quantization-by-ml_amd/`` imports it.

Restates, on the CPU and in a fixed arithmetic order, what the reference computes on the hot path
(citations relative to /root/reference):

* ``nearest``            -- ``Codebook.forward`` search core, ``vector_quantization/codebooks.py:386-391``
                            (``-cdist`` / einsum, then ``gumbel_sample`` deterministic branch
                            ``utils/general.py:126-136``: first argmax).
* ``vq_forward``         -- ``codebooks.py:393-397`` (quantize == ``codebook[idx]`` in both train and eval),
                            ``vector_quantize_pytorch.py:261-279`` (straight-through ``x + (q - x)``) and
                            ``:337,361-364`` (commitment loss = mean((q - x)^2) over ALL heads/elements).
* ``rvq_forward``        -- ``residual_vq.py:154-155,212-243``: ``residual -= quantized`` /
                            ``quantized_out = 0.0 + q1 + q2 ...`` with the value the layer RETURNS
                            (the straight-through expression in train mode).

All floating point is numpy float32 element-wise arithmetic (one IEEE rounding per operation, no
contraction), the search itself is the C k-ordered fmaf chain.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

EUCLID = 0
DOT = 1

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libvq_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    """Compile vq_oracle.c with gcc (oracle/Makefile).  Returns the .so path."""
    src = os.path.join(_HERE, "vq_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, "-s"], check=True)
    return _SO


def _load():
    global _lib
    if _lib is None:
        build()
        lib = ctypes.CDLL(_SO)
        i64, f32p, i64p = ctypes.c_int64, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int64)
        lib.vq_oracle_nearest_f32.argtypes = [f32p, i64, i64, f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, i64p, f32p]
        lib.vq_oracle_nearest_f32.restype = ctypes.c_int
        lib.vq_oracle_similarities_f32.argtypes = [f32p, i64, i64, f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, f32p]
        lib.vq_oracle_similarities_f32.restype = ctypes.c_int
        lib.vq_oracle_num_threads.restype = ctypes.c_int
        _lib = lib
    return _lib


def num_threads() -> int:
    return int(_load().vq_oracle_num_threads())


def _f32p(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def nearest(x: np.ndarray, cb: np.ndarray, metric: int = EUCLID):
    """x [M, D] f32, cb [K, D] f32 -> (idx int64 [M], best f32 [M])."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    cb = np.ascontiguousarray(cb, dtype=np.float32)
    assert x.ndim == 2 and cb.ndim == 2 and x.shape[1] == cb.shape[1]
    M, D = x.shape
    K = cb.shape[0]
    idx = np.empty(M, dtype=np.int64)
    best = np.empty(M, dtype=np.float32)
    rc = _load().vq_oracle_nearest_f32(_f32p(x), D, M, _f32p(cb), K, D, metric,
                                       idx.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), _f32p(best))
    if rc != 0:
        raise RuntimeError("vq_oracle_nearest_f32 failed")
    return idx, best


def similarities(x: np.ndarray, cb: np.ndarray, metric: int = EUCLID) -> np.ndarray:
    x = np.ascontiguousarray(x, dtype=np.float32)
    cb = np.ascontiguousarray(cb, dtype=np.float32)
    M, D = x.shape
    K = cb.shape[0]
    sim = np.empty((M, K), dtype=np.float32)
    rc = _load().vq_oracle_similarities_f32(_f32p(x), D, M, _f32p(cb), K, D, metric, _f32p(sim))
    if rc != 0:
        raise RuntimeError("vq_oracle_similarities_f32 failed")
    return sim


def vq_forward(x: np.ndarray, cb: np.ndarray, metric: int = EUCLID, training: bool = False):
    """One ``Codebook`` + quantize step on flattened input.

    x [H, M, D], cb [H, K, D]  ->  dict(out [H,M,D], idx [H,M] int64, best [H,M], sq_err float64 scalar,
    q [H,M,D]).  ``out`` is ``q`` in eval and the straight-through value ``x + (q - x)`` in training;
    ``sq_err`` is sum((q - x)^2) over everything (commitment loss = weight * sq_err / x.size).
    """
    x = np.ascontiguousarray(x, dtype=np.float32)
    cb = np.ascontiguousarray(cb, dtype=np.float32)
    H, M, D = x.shape
    idx = np.empty((H, M), dtype=np.int64)
    best = np.empty((H, M), dtype=np.float32)
    q = np.empty_like(x)
    for h in range(H):
        idx[h], best[h] = nearest(x[h], cb[h], metric)
        q[h] = cb[h][idx[h]]
    diff = q - x
    sq_err = float(np.sum(diff.astype(np.float64) ** 2))
    out = (x + diff) if training else q
    return dict(out=out, idx=idx, best=best, sq_err=sq_err, q=q)


def rvq_forward(x: np.ndarray, cbs: np.ndarray, metric: int = EUCLID, training: bool = False):
    """ResidualVQ loop.  x [M, D], cbs [Q, K, D] -> dict(out [M,D], idx [M,Q], best [M,Q], sq_err [Q])."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    cbs = np.ascontiguousarray(cbs, dtype=np.float32)
    M, D = x.shape
    Q = cbs.shape[0]
    residual = x.copy()
    out = np.zeros_like(x)  # reference starts from python 0.0: 0.0 + q1 == q1 (and -0.0 -> +0.0)
    idx = np.empty((M, Q), dtype=np.int64)
    best = np.empty((M, Q), dtype=np.float32)
    sq_err = np.empty(Q, dtype=np.float64)
    for s in range(Q):
        i, b = nearest(residual, cbs[s], metric)
        q = cbs[s][i]
        diff = q - residual
        sq_err[s] = float(np.sum(diff.astype(np.float64) ** 2))
        quantized = (residual + diff) if training else q
        residual = residual - quantized
        out = out + quantized
        idx[:, s] = i
        best[:, s] = b
    return dict(out=out, idx=idx, best=best, sq_err=sq_err)


def pack_key(best: np.ndarray, idx: np.ndarray, metric: int = EUCLID) -> np.ndarray:
    """(value, index) -> one SIGNED int64 whose MIN picks the winner with lowest-index ties.

    hi word: m = "smaller is better" unsigned image of the value, stored with its top bit flipped so that signed
    comparison == unsigned comparison of m.  Euclid: 1 + the IEEE bits of the non-negative sqrt distance, and 0 for a
    NaN (argmax treats a NaN similarity as the maximum, utils/general.py:128: it must beat distance 0); dot: complement
    of the usual order-preserving float->uint map, NaN canonicalised to the positive quiet NaN (which that map already
    places above +inf).
    lo word: the code index.
    """
    best = np.ascontiguousarray(best, dtype=np.float32)
    nan = np.isnan(best)
    bits = best.view(np.uint32)
    if metric == DOT:
        bits = np.where(nan, np.uint32(0x7FC00000), bits).astype(np.uint32)
        neg = (bits >> np.uint32(31)) != 0
        mono = np.where(neg, ~bits, bits | np.uint32(0x80000000)).astype(np.uint32)
        m = ~mono
    else:
        m = np.where(nan, np.uint32(0), bits + np.uint32(1)).astype(np.uint32)
    hi = (m ^ np.uint32(0x80000000)).view(np.int32).astype(np.int64)
    return (hi << np.int64(32)) | idx.astype(np.int64)


def unpack_key(key: np.ndarray, metric: int = EUCLID):
    """Inverse of pack_key -> (best f32, idx int64)."""
    key = np.ascontiguousarray(key, dtype=np.int64)
    idx = key & np.int64(0xFFFFFFFF)
    m = ((key >> np.int64(32)).astype(np.int32).view(np.uint32)) ^ np.uint32(0x80000000)
    if metric == DOT:
        mono = ~m
        pos = (mono >> np.uint32(31)) != 0
        bits = np.where(pos, mono ^ np.uint32(0x80000000), ~mono).astype(np.uint32)
    else:
        bits = np.where(m == 0, np.uint32(0x7FC00000), m - np.uint32(1)).astype(np.uint32)
    return bits.view(np.float32), idx
