"""GPU: non-finite inputs on every launch path.

ATen's argmax treats NaN as the maximum and returns the FIRST one (reference: utils/general.py:128 over
codebooks.py:128-129,386): a row holding a NaN answers code 0, a row holding +-inf the first code whose distance chain meets
inf - inf, a code holding a NaN wins EVERY row, a code holding an inf the rows on one side of it.  The oracle restates that
rule and is pinned to the reference by the nf_* fixtures (tests/test_oracle_golden.py, the module tests); here every
kernel / launch plan must reproduce the oracle bit for bit on poisoned inputs: the one-block kernel at every padded
dim, the persistent kernel, the wave-pair kernel, K split into packed keys, the main + tail plan, rows wider than 512
dims (keys and fused), 2-byte rows, residual stacks, shard keys, the log-sum-exp variant and the similarity emission.
"""
from __future__ import annotations

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
NAN, INF = float("nan"), float("inf")


def _native():
    from vector_quantization import native

    native.load()
    return native


def poison_rows(x: torch.Tensor, seed: int = 0):
    """x [M, D] (in place): a NaN element, a +inf, a -inf, an all-inf row, +inf and -inf together, a NaN in the last dim, spread
    over the row blocks (first / middle / last rows).  Returns the poisoned row numbers."""
    M, D = x.shape
    g = np.random.default_rng(seed)
    rows = sorted(set([0, 1, 2, 3, 4, min(5, M - 1), M // 2, M - 1] + [int(v) for v in g.integers(0, M, 12)]))
    kinds = ["nan", "inf", "-inf", "allinf", "mixed", "nanlast", "nan", "-inf"]
    for i, r in enumerate(rows):
        kind = kinds[i % len(kinds)]
        d = int(g.integers(0, D))
        if kind == "nan":
            x[r, d] = NAN
        elif kind == "inf":
            x[r, d] = INF
        elif kind == "-inf":
            x[r, d] = -INF
        elif kind == "allinf":
            x[r, :] = INF
        elif kind == "mixed":
            x[r, 0] = INF
            x[r, D - 1] = -INF if D > 1 else INF
        else:
            x[r, D - 1] = NAN
    return rows


def same_values(got: np.ndarray, want: np.ndarray) -> bool:
    """bit-equal where finite / inf, NaN where NaN (the payload of a NaN is not part of the contract)"""
    gn, wn = np.isnan(got), np.isnan(want)
    return bool(np.array_equal(gn, wn) and np.array_equal(got[~gn].view(np.uint32), want[~wn].view(np.uint32)))


def check_rows(oracle, r, x, cb, metric, rows=None):
    """native result `r` of a single-stage search of x [M, D] against cb [K, D], on `rows` (default: all)."""
    idx = r["idx"][0, :, 0].cpu().numpy()
    best = r["best"][0, :, 0].cpu().numpy()
    sel = np.arange(x.shape[0]) if rows is None else np.asarray(sorted(set(rows)))
    ri, rb = oracle.nearest(x[sel].numpy(), cb.numpy(), metric)
    np.testing.assert_array_equal(idx[sel], ri)
    assert same_values(best[sel], rb), "winning values differ from the oracle"
    if r.get("out") is not None:
        out = r["out"][0].cpu().numpy()
        want = cb.numpy()[ri]
        assert np.array_equal(np.isnan(out[sel]), np.isnan(want)) and np.array_equal(np.nan_to_num(out[sel]), np.nan_to_num(want))


# (M, K, D): which kernel / plan the launcher picks on a 256-CU device
SINGLE = [
    (3000, 300, 24),        # Dp = 32, one-block kernel
    (3000, 300, 64),        # Dp = 64
    (2500, 1000, 100),      # Dp = 128
    (3000, 500, 256),       # Dp = 256, short sweep (not persistent)
    (131072 + 77, 1024, 256),   # persistent kernel (two row blocks per CU) ... + main / tail plan
    (70000, 1024, 256),     # whole rounds fused + K-split tail
    (40000, 4096, 256),     # K split over workgroups (keys + finalize)
    (300, 4096, 128),       # few rows: K split until the chip is full
    (33000, 300, 320),      # wave-pair kernel (256 < D <= 512), fused
    (600, 8192, 512),       # wave-pair kernel in keys mode
    (200, 150, 700),        # wide rows: three slices, keys path
    (66000, 128, 768),      # wide rows: the last slice finishes the inference call itself
]


@pytest.mark.parametrize("M,K,D", SINGLE)
@pytest.mark.parametrize("metric", [0, 1])
def test_poisoned_rows_every_launch_path(oracle, M, K, D, metric):
    native = _native()
    g = torch.Generator().manual_seed(M + K + D)
    x = torch.randn((M, D), generator=g)
    cb = torch.randn((K, D), generator=g)
    rows = poison_rows(x, seed=M)
    xd, cbd = x.to(DEV)[None], cb.to(DEV)[None, None].contiguous()
    r = native.quantize(xd, cbd, metric=metric, want_best=True)
    torch.cuda.synchronize()
    sample = rows + list(range(min(M, 300))) + list(range(max(0, M - 300), M))
    check_rows(oracle, r, x, cb, metric, None if M <= 4000 else sample)
    if M > 4000:  # every row against the one-thread-per-row kernel (itself checked against the oracle at the small sizes)
        s = native.quantize(xd, cbd, metric=metric, want_best=True, flags=native.F_FORCE_SIMPLE)
        assert torch.equal(r["idx"], s["idx"])
        assert same_values(r["best"].cpu().numpy().ravel(), s["best"].cpu().numpy().ravel())


@pytest.mark.parametrize("M,K,D", [(2000, 300, 64), (3000, 500, 256), (131072 + 77, 1024, 256), (20000, 300, 320), (300, 4096, 128),
                                   (200, 150, 700), (66000, 128, 768)])
@pytest.mark.parametrize("metric", [0, 1])
@pytest.mark.parametrize("what", ["nan", "inf"])
def test_poisoned_codebook_every_launch_path(oracle, M, K, D, metric, what):
    """A code holding a NaN wins EVERY row (the lower of two such codes); a code holding an inf wins the rows on one side."""
    native = _native()
    g = torch.Generator().manual_seed(M + K + D + 1)
    x = torch.randn((M, D), generator=g)
    cb = torch.randn((K, D), generator=g)
    if what == "nan":
        cb[K - 3, D // 2] = NAN
        cb[K // 3, 0] = NAN
    else:
        cb[K // 3, D - 1] = INF
        cb[K - 1, 0] = -INF
    x[1, 3] = NAN  # and a poisoned row on top
    xd, cbd = x.to(DEV)[None], cb.to(DEV)[None, None].contiguous()
    r = native.quantize(xd, cbd, metric=metric, want_best=True)
    torch.cuda.synchronize()
    sample = list(range(min(M, 400))) + list(range(max(0, M - 200), M))
    check_rows(oracle, r, x, cb, metric, None if M <= 4000 else sample)
    if what == "nan":
        assert bool((r["idx"][0, 2:, 0] == K // 3).all()), "every finite row must pick the first code that holds a NaN"
    if M > 4000:
        s = native.quantize(xd, cbd, metric=metric, want_best=True, flags=native.F_FORCE_SIMPLE)
        assert torch.equal(r["idx"], s["idx"])


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M,K,D", [(3000, 300, 64), (3000, 500, 256), (20000, 300, 320)])
def test_poisoned_two_byte_rows(oracle, dtype, M, K, D):
    native = _native()
    g = torch.Generator().manual_seed(5)
    x = torch.randn((M, D), generator=g).to(dtype)
    cb = torch.randn((K, D), generator=g)
    xf = x.float()
    rows = poison_rows(xf, seed=3)
    x = xf.to(dtype)  # NaN / inf survive the narrowing
    r = native.quantize(x.to(DEV)[None], cb.to(DEV)[None, None].contiguous(), want_best=True)
    torch.cuda.synchronize()
    check_rows(oracle, r, x.float(), cb, 0, rows + list(range(300)))


@pytest.mark.parametrize("D,K,Q", [(64, 128, 3), (256, 256, 4), (100, 70, 2), (512, 64, 2)])
@pytest.mark.parametrize("ste", [False, True])
@pytest.mark.parametrize("poison", ["rows", "codes"])
def test_poisoned_residual_stack(oracle, D, K, Q, ste, poison, residual_plan):
    """One fused launch for all stages: a repaired winner must also feed the residual of the following stages."""
    from gen import make_rvq_codebooks

    native = _native()
    M = 1500
    g = torch.Generator().manual_seed(D + K + Q)
    x = torch.randn((M, D), generator=g)
    cbs = make_rvq_codebooks(Q, K, D, "S")
    if poison == "rows":
        poison_rows(x, seed=Q)
    else:
        cbs[1, K // 2, D // 3] = NAN
        cbs[Q - 1, 5, 0] = INF
    with np.errstate(invalid="ignore"):
        ref = oracle.rvq_forward(x.numpy(), cbs.numpy(), oracle.EUCLID, training=ste)
    r = native.quantize(x.to(DEV)[None], cbs.to(DEV)[None].contiguous(), ste=ste, want_best=True, want_sq_err=ste)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(r["idx"][0].cpu().numpy(), ref["idx"])
    assert same_values(r["best"][0].cpu().numpy().ravel(), ref["best"].ravel())
    out = r["out"][0].cpu().numpy()
    assert np.array_equal(np.isnan(out), np.isnan(ref["out"]))
    np.testing.assert_array_equal(np.nan_to_num(out, nan=0.0, posinf=1e30, neginf=-1e30),
                                  np.nan_to_num(ref["out"], nan=0.0, posinf=1e30, neginf=-1e30))
    if ste:
        got, want = r["sq_err"].cpu().numpy(), ref["sq_err"]
        assert np.array_equal(np.isfinite(got), np.isfinite(want)), (got, want)


@pytest.mark.parametrize("metric", [0, 1])
@pytest.mark.parametrize("D", [64, 512, 700])
def test_shard_keys_with_nonfinite_values(oracle, metric, D):
    """Packed keys of K shards: a NaN similarity is the best key, the lowest GLOBAL index wins among NaNs -- the MIN over the
    shards equals the full-codebook rule (what the sharded multi-GPU path reduces with RCCL)."""
    native = _native()
    M, K, G = 700, 512, 4
    g = torch.Generator().manual_seed(D)
    x = torch.randn((M, D), generator=g)
    cb = torch.randn((K, D), generator=g)
    rows = poison_rows(x, seed=9)
    cb[3 * K // 4 + 5, 1] = NAN     # lives in shard 3
    cb[K // 4 + 7, 0] = NAN         # and in shard 1: the lower one wins every finite row
    xd = x.to(DEV)[None]
    keys = torch.empty((1, M), dtype=torch.int64, device=DEV)
    native.keys_init(keys)
    per = K // G
    for s in range(G):
        native.search_keys(xd, cb[s * per:(s + 1) * per].to(DEV)[None].contiguous(), keys, metric=metric, idx_offset=s * per)
    torch.cuda.synchronize()
    ri, rb = oracle.nearest(x.numpy(), cb.numpy(), metric)
    best, idx = oracle.unpack_key(keys[0].cpu().numpy(), metric)
    np.testing.assert_array_equal(idx, ri)
    assert same_values(best, rb)
    fin = native.finalize_keys(xd, cb.to(DEV)[None].contiguous(), keys, metric=metric)
    np.testing.assert_array_equal(fin["idx"][0].cpu().numpy(), ri)
    assert same_values(fin["best"][0].cpu().numpy(), rb)
    assert len(rows) > 0


def test_log_sum_exp_of_poisoned_rows_is_nan():
    native = _native()
    M, K, D = 2000, 256, 64
    g = torch.Generator().manual_seed(1)
    x = torch.randn((M, D), generator=g)
    cb = torch.randn((K, D), generator=g)
    x[5, 7] = NAN
    x[900, 0] = NAN
    r = native.quantize(x.to(DEV)[None], cb.to(DEV)[None, None].contiguous(), want_lse=True, want_best=True)
    torch.cuda.synchronize()
    lse = r["lse"][0].cpu()
    want = torch.logsumexp(-torch.cdist(x, cb), dim=-1)
    assert torch.equal(torch.isnan(lse), torch.isnan(want))
    ok = ~torch.isnan(want)
    torch.testing.assert_close(lse[ok], want[ok], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("D", [64, 256, 700])
@pytest.mark.parametrize("metric", [0, 1])
def test_similarities_keep_nan(oracle, D, metric):
    native = _native()
    M, K = 300, 200
    g = torch.Generator().manual_seed(2)
    x = torch.randn((M, D), generator=g)
    cb = torch.randn((K, D), generator=g)
    poison_rows(x, seed=4)
    cb[17, 3] = NAN
    sims = native.similarities(x.to(DEV)[None], cb.to(DEV)[None].contiguous(), metric=metric)[0].cpu().numpy()
    want = oracle.similarities(x.numpy(), cb.numpy(), metric)
    assert np.array_equal(np.isnan(sims), np.isnan(want))
    assert np.array_equal(sims[~np.isnan(want)].view(np.uint32), want[~np.isnan(want)].view(np.uint32))
