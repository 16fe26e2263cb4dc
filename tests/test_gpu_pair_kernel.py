"""GPU: the wave-pair search kernel (vq_search_pair512, 256 < D <= 512) on its FUSED path -- enough rows that the launcher
does not split K -- at ragged sizes: D not a multiple of 4 / 8 / 256, K not a multiple of 32, a partial last row block,
straight-through + squared error, the LSE variant, 2-byte rows, both metrics.  Checked bit for bit against the CPU oracle on
a row sample and against the scalar kernel (same k-ordered chain, one thread per row) on every row; the old one-wave kernel
(VQ_SINGLE_WAVE_512) is the third witness where the environment switch is honoured (fresh process)."""
from __future__ import annotations

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _native():
    from vector_quantization import native

    native.load()
    return native


CASES = [
    # (H, M, K, D, metric)
    (1, 33000, 1000, 512, 0),
    (1, 33111, 777, 300, 0),
    (2, 20000, 100, 384, 1),
    (1, 40001, 33, 511, 0),
    (1, 36000, 2048, 260, 0),
    (1, 33000, 64, 257, 1),
]


@pytest.mark.parametrize("H,M,K,D,metric", CASES)
@pytest.mark.parametrize("training", [False, True])
def test_fused_pair_kernel_bit_exact(oracle, H, M, K, D, metric, training):
    native = _native()
    g = torch.Generator().manual_seed(M + K + D)
    x = torch.randn((H, M, D), generator=g).to(DEV)
    cb = torch.randn((H, 1, K, D), generator=g).to(DEV)
    r = native.quantize(x, cb, metric=metric, ste=training, want_sq_err=training)
    s = native.quantize(x, cb, metric=metric, ste=training, want_sq_err=training, flags=native.F_FORCE_SIMPLE)
    assert torch.equal(r["idx"], s["idx"])
    assert torch.equal(r["best"].view(torch.int32), s["best"].view(torch.int32))
    assert torch.equal(r["out"], s["out"])
    if training:
        torch.testing.assert_close(r["sq_err"], s["sq_err"], rtol=1e-6, atol=0)
    rows = torch.randperm(M, generator=torch.Generator().manual_seed(1))[:400]
    rows = torch.cat([rows, torch.arange(M - 40, M)])  # incl. the partial last row block
    for h in range(H):
        ri, rb = oracle.nearest(x[h, rows].cpu().numpy(), cb[h, 0].cpu().numpy(), metric)
        np.testing.assert_array_equal(r["idx"][h, rows, 0].cpu().numpy(), ri)
        assert np.array_equal(r["best"][h, rows, 0].cpu().numpy().view(np.uint32), rb.view(np.uint32))
    hh = torch.arange(H, device=DEV)[:, None]
    gathered = cb[:, 0][hh, r["idx"][..., 0]]
    want = x + (gathered - x) if training else gathered
    assert torch.equal(r["out"], want)


def test_pair_kernel_lse_and_half_rows(oracle):
    native = _native()
    H, M, K, D = 1, 33000, 500, 320
    g = torch.Generator().manual_seed(5)
    x = torch.randn((H, M, D), generator=g).to(DEV)
    cb = torch.randn((H, 1, K, D), generator=g).to(DEV)
    base = native.quantize(x, cb)
    # log-sum-exp from the same sweep
    r = native.quantize(x, cb, want_lse=True)
    assert torch.equal(r["idx"], base["idx"]) and torch.equal(r["out"], base["out"])
    sims = native.similarities(x[:, :2000].contiguous(), cb[:, 0])
    torch.testing.assert_close(r["lse"][:, :2000], torch.logsumexp(sims.double(), -1).float(), rtol=2e-5, atol=2e-5)
    # bf16 / fp16 rows are widened in the prologue: identical to searching the widened copy
    for dt in (torch.bfloat16, torch.float16):
        xh = x.to(dt)
        a = native.quantize(xh, cb)
        b = native.quantize(xh.float(), cb)
        assert torch.equal(a["idx"], b["idx"]) and torch.equal(a["best"].view(torch.int32), b["best"].view(torch.int32))


def test_pair_kernel_ties_take_the_lowest_index(oracle):
    """Exact-grid values with the second half of the codebook duplicating the first: every winner must come from the first
    half (the tie rule runs in wave B on accumulators that started in wave A)."""
    native = _native()
    M, K, D = 33000, 512, 512
    g = torch.Generator().manual_seed(9)
    x = (torch.randint(-16, 17, (1, M, D), generator=g).float() / 8.0).to(DEV)
    half = torch.randint(-16, 17, (1, 1, K // 2, D), generator=g).float() / 8.0
    cb = torch.cat([half, half], dim=2).to(DEV)
    r = native.quantize(x, cb)
    assert int(r["idx"].max()) < K // 2
    rows = torch.arange(0, M, 97)
    ri, rb = oracle.nearest(x[0, rows].cpu().numpy(), cb[0, 0].cpu().numpy(), 0)
    np.testing.assert_array_equal(r["idx"][0, rows, 0].cpu().numpy(), ri)
    assert np.array_equal(r["best"][0, rows, 0].cpu().numpy().view(np.uint32), rb.view(np.uint32))


def _same_bits(a: torch.Tensor, b: torch.Tensor) -> bool:
    an, bn = torch.isnan(a), torch.isnan(b)
    z = torch.zeros_like(a)
    return bool(torch.equal(an, bn)) and bool(torch.equal(torch.where(an, z, a).view(torch.int32), torch.where(bn, z, b).view(torch.int32)))


RESIDUAL_CASES = [
    # (H, Q, M, K, D, metric)     K >= 256: at least 8 tiles per sweep, else the launcher keeps the one-wave kernel
    (1, 4, 300, 256, 512, 0),
    (1, 3, 1000, 1000, 300, 0),
    (1, 8, 129, 260, 384, 0),
    (2, 2, 200, 512, 500, 0),
    (1, 5, 4097, 288, 257, 0),
    (1, 3, 500, 300, 512, 1),
    (1, 20, 64, 256, 272, 0),     # the largest stack with squared errors the one-wave kernel (the witness) holds
]


@pytest.mark.parametrize("H,Q,M,K,D,metric", RESIDUAL_CASES)
@pytest.mark.parametrize("training", [False, True])
def test_residual_stacks_on_the_pair_kernel(oracle, H, Q, M, K, D, metric, training, residual_plan):
    """256 < D <= 512, Q > 1 (round 3): both waves of a pair update their half of the residual after every sweep.  Bit-exact
    against the CPU oracle's residual loop (residual_vq.py:212-243) and equal to the one-wave kernel (VQ_PAIR_NO_MULTI=1)."""
    import os
    native = _native()
    g = torch.Generator().manual_seed(Q * 1000 + M + K + D)
    x = torch.randn((H, M, D), generator=g)
    cbs = torch.stack([torch.stack([torch.randn((K, D), generator=g) * 2.0 ** (-i / 2.0) for i in range(Q)]) for _ in range(H)])
    xd, cd = x.to(DEV), cbs.to(DEV)
    r = native.quantize(xd, cd, metric=metric, ste=training, want_sq_err=training)
    os.environ["VQ_PAIR_NO_MULTI"] = "1"
    try:
        o = native.quantize(xd, cd, metric=metric, ste=training, want_sq_err=training)
    finally:
        os.environ.pop("VQ_PAIR_NO_MULTI", None)
    torch.cuda.synchronize()
    assert torch.equal(r["idx"], o["idx"])
    assert _same_bits(r["best"], o["best"]) and _same_bits(r["out"], o["out"])
    if training:
        torch.testing.assert_close(r["sq_err"], o["sq_err"], rtol=1e-6, atol=0)
    for h in range(H):
        ref = oracle.rvq_forward(x[h].numpy(), cbs[h].numpy(), metric, training=training)
        np.testing.assert_array_equal(r["idx"][h].cpu().numpy(), ref["idx"])
        assert np.array_equal(r["best"][h].cpu().numpy().view(np.uint32), ref["best"].view(np.uint32))
        np.testing.assert_array_equal(r["out"][h].cpu().numpy(), ref["out"])
    if training and H == 1:
        np.testing.assert_allclose(r["sq_err"].cpu().numpy().reshape(-1), ref["sq_err"], rtol=1e-6)


@pytest.mark.parametrize("training", [False, True])
@pytest.mark.parametrize("poison_codes", [False, True])
def test_residual_stacks_on_the_pair_kernel_non_finite(oracle, training, poison_codes, residual_plan):
    """Rows holding NaN / inf (and a NaN inside a later stage's codebook): wave A dumps its half of the flagged rows, wave B
    repairs them by the plain rule (ATen's argmax: the first NaN wins) -- equal to the oracle's residual loop."""
    native = _native()
    Q, M, K, D = 3, 700, 320, 400
    g = torch.Generator().manual_seed(77)
    x = torch.randn((M, D), generator=g)
    cbs = torch.stack([torch.randn((K, D), generator=g) * 2.0 ** (-i / 2.0) for i in range(Q)])
    x[5, 3] = float("nan")
    x[64, 399] = float("inf")
    x[699, 256] = float("-inf")
    x[130, 0] = float("nan")
    x[130, 300] = float("inf")
    if poison_codes:
        cbs[1, 17, 290] = float("nan")
    with np.errstate(invalid="ignore", over="ignore"):
        ref = oracle.rvq_forward(x.numpy(), cbs.numpy(), 0, training=training)
    r = native.quantize(x[None].to(DEV), cbs[None].contiguous().to(DEV), ste=training, want_sq_err=training)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(r["idx"][0].cpu().numpy(), ref["idx"])
    assert _same_bits(r["best"][0].cpu(), torch.from_numpy(ref["best"]))
    assert _same_bits(r["out"][0].cpu(), torch.from_numpy(ref["out"]))
