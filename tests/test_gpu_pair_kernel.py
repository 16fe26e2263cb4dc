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
