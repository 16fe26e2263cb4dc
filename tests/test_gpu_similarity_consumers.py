"""GPU parity for the similarity-consuming entry points.

* vq_similarities_f32: the [H, M, K] matrix must be BIT-identical to the oracle's (same k-ordered fmaf chain,
  correctly rounded sqrt) -- it is the third return value of the reference's Codebook.forward.
* vq_softmax_stats_f32: log-sum-exp and target logit of scale * similarity, against float64 arithmetic on the
  oracle's similarities.  Tolerance 2e-6 relative / 2e-5 absolute (native sqrt / exp / log inside the kernel).
"""
from __future__ import annotations

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from gen import make_codebook, make_x  # noqa: E402


def _native():
    from vector_quantization import native

    native.load()
    return native


SHAPES = [
    # (H, M, K, D)
    (1, 512, 1024, 256),
    (1, 300, 256, 64),
    (4, 130, 520, 64),
    (1, 111, 301, 100),   # K % 4 != 0 -> scalar stores, D padded
    (2, 33, 7, 5),
    (1, 40, 1, 16),
    (1, 70, 1000, 128),
    (1, 64, 2048, 512),
    (1, 100, 96, 32),
    (1, 50, 40, 600),     # D > 512 -> scalar kernel
]


def _oracle_sims(oracle, x, cb, metric):
    return np.stack([oracle.similarities(x[h].numpy(), cb[h].numpy(), metric) for h in range(x.shape[0])])


@pytest.mark.parametrize("H,M,K,D", SHAPES)
@pytest.mark.parametrize("cls", ["S", "G"])
@pytest.mark.parametrize("metric", [0, 1])
def test_similarities_bit_exact(oracle, H, M, K, D, cls, metric):
    native = _native()
    x, cb = make_x((H, M, D), cls), make_codebook(H, K, D, cls)
    want = _oracle_sims(oracle, x, cb, metric)
    got = native.similarities(x.cuda(), cb.cuda(), metric=metric)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(got.cpu().numpy(), want)


@pytest.mark.parametrize("metric", [0, 1])
def test_similarities_scalar_kernel_agrees(oracle, metric):
    native = _native()
    x, cb = make_x((2, 100, 48), "S"), make_codebook(2, 333, 48, "S")
    a = native.similarities(x.cuda(), cb.cuda(), metric=metric)
    b = native.similarities(x.cuda(), cb.cuda(), metric=metric, flags=native.F_FORCE_SIMPLE)
    torch.cuda.synchronize()
    assert torch.equal(a, b)


def test_similarities_strided_rows_and_chunk_views(oracle):
    """Head-split views (rows strided by heads * d) and writing into a row-chunk of a larger buffer."""
    native = _native()
    rows, heads, d, K = 96, 3, 32, 64
    x4 = make_x((rows, heads, d), "S")
    cb = make_codebook(heads, K, d, "S")
    flat = x4.cuda().permute(1, 0, 2)  # [h, rows, d] strided
    big = torch.full((heads, rows + 10, K), 7.0, device="cuda")
    native.similarities(flat, cb.cuda(), out=big[:, 5:5 + rows])
    torch.cuda.synchronize()
    want = _oracle_sims(oracle, x4.permute(1, 0, 2).contiguous(), cb, 0)
    np.testing.assert_array_equal(big[:, 5:5 + rows].cpu().numpy(), want)
    assert bool((big[:, :5] == 7.0).all()) and bool((big[:, 5 + rows:] == 7.0).all())


def test_similarities_max_is_the_search_winner(oracle):
    """argmax of the emitted matrix (first index) == the search kernel's index, max == -best, bitwise."""
    native = _native()
    x, cb = make_x((1, 2048, 256), "R"), make_codebook(1, 1024, 256, "R")
    xs, cbs = x.cuda(), cb.cuda()
    sims = native.similarities(xs, cbs)
    r = native.quantize(xs, cbs[:, None].contiguous())
    torch.cuda.synchronize()
    assert torch.equal(sims.argmax(-1), r["idx"][..., 0])
    assert torch.equal(sims.max(-1).values, -r["best"][..., 0])


@pytest.mark.parametrize("H,M,K,D", [s for s in SHAPES if s[3] <= 512])
@pytest.mark.parametrize("metric,scale", [(0, 1.0), (1, 1.0), (0, -100.0), (1, 0.25)])
def test_softmax_stats(oracle, H, M, K, D, metric, scale):
    native = _native()
    x, cb = make_x((H, M, D), "S"), make_codebook(H, K, D, "S")
    sims = _oracle_sims(oracle, x, cb, metric).astype(np.float64)
    logits = scale * sims
    mx = logits.max(-1, keepdims=True)
    lse = (mx + np.log(np.exp(logits - mx).sum(-1, keepdims=True)))[..., 0]
    g = torch.Generator().manual_seed(5)
    target = torch.randint(0, K, (H, M), generator=g)
    target[:, ::7] = -1  # ignored rows
    got_lse, got_tl = native.softmax_stats(x.cuda(), cb.cuda(), metric=metric, scale=scale, target=target.cuda())
    torch.cuda.synchronize()
    np.testing.assert_allclose(got_lse.cpu().numpy(), lse, rtol=2e-6, atol=2e-5)
    t = target.numpy()
    want_tl = np.where(t >= 0, np.take_along_axis(logits, np.maximum(t, 0)[..., None], -1)[..., 0], 0.0)
    np.testing.assert_allclose(got_tl.cpu().numpy(), want_tl, rtol=2e-6, atol=2e-5)
    # lse only
    lse2, none = native.softmax_stats(x.cuda(), cb.cuda(), metric=metric, scale=scale)
    torch.cuda.synchronize()
    assert none is None and torch.equal(lse2, got_lse)


def test_softmax_stats_cross_entropy_full_size():
    """cfg2-sized rows: mean(lse - target_logit) equals F.cross_entropy on chunks of the emitted similarities."""
    native = _native()
    M, K, D = 65536, 1024, 256
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn((1, M, D), device="cuda", generator=g)
    cb = torch.randn((1, K, D), device="cuda", generator=g)
    target = torch.randint(0, K, (1, M), device="cuda", generator=g)
    lse, tl = native.softmax_stats(x, cb, target=target)
    ce = float((lse - tl).double().mean())
    acc = 0.0
    for r0 in range(0, M, 8192):
        sims = native.similarities(x[:, r0:r0 + 8192], cb)
        acc += float(torch.nn.functional.cross_entropy(sims[0].double(), target[0, r0:r0 + 8192], reduction="sum"))
    assert abs(ce - acc / M) <= 1e-5 * max(1.0, abs(ce))


def _ce_grad_reference(x, cb, target, metric, coef):
    """float64 autograd of coef * sum_valid (lse - logit[target]) on the CPU."""
    xd = x.double().clone().requires_grad_(True)
    cd = cb.double()
    sims = -torch.cdist(xd, cd) if metric == 0 else xd @ cd.transpose(-1, -2)
    ce = torch.nn.functional.cross_entropy(sims.reshape(-1, sims.shape[-1]), target.reshape(-1), ignore_index=-1,
                                           reduction="sum")
    (coef * ce).backward()
    return xd.grad


@pytest.mark.parametrize("H,M,K,D", [s for s in SHAPES if s[3] <= 512] + [(2, 150, 300, 400)])
@pytest.mark.parametrize("metric", [0, 1])
def test_fused_cross_entropy_backward(H, M, K, D, metric):
    """vq_ce_backward_f32 against float64 autograd; tolerance 2e-5 of the largest gradient entry."""
    native = _native()
    x, cb = make_x((H, M, D), "S"), make_codebook(H, K, D, "S")
    if metric == 1:
        x = x * 0.25  # keep the dot-product softmax away from one-hot saturation
    g = torch.Generator().manual_seed(11)
    target = torch.randint(0, K, (H, M), generator=g)
    target[:, ::5] = -1
    # make one row coincide with a code: dist == 0 -> subgradient 0 for that code (ATen masks res == 0)
    if metric == 0 and M > 3:
        x[0, 3] = cb[0, K // 2]
    coef = 0.37
    want = _ce_grad_reference(x, cb, target, metric, coef)
    xs, cbs, ts = x.cuda(), cb.cuda(), target.cuda()
    lse, tl = native.softmax_stats(xs, cbs, metric=metric, target=ts)
    got = native.ce_backward(xs, cbs, lse, tl, ts, torch.tensor([coef], device="cuda"), metric=metric)
    torch.cuda.synchronize()
    scale = max(float(want.abs().max()), coef)  # K == 1: the exact gradient is 0, fp32 leaves 1e-7 of noise
    np.testing.assert_allclose(got.cpu().double().numpy(), want.numpy(), atol=2e-5 * scale, rtol=2e-4)
    assert bool((got[:, ::5] == 0).all()), "ignored rows must get exactly zero gradient"


@pytest.mark.parametrize("H,M,K,D", [(1, 128, 32, 256), (1, 1000, 1000, 256), (2, 333, 1030, 200), (1, 4097, 64, 132),
                                     (3, 129, 7, 256), (1, 70000, 96, 256),
                                     # Dp = 512: four roles per row block (S1, S2, G1, G2), 64 rows per workgroup
                                     (1, 128, 64, 512), (2, 333, 1030, 400), (1, 4097, 96, 300), (1, 65, 33, 512), (1, 20000, 260, 512)])
@pytest.mark.parametrize("metric", [0, 1])
def test_cross_entropy_backward_wave_pair_kernel_equals_one_wave_kernel(H, M, K, D, metric):
    """Dp = 256 / 512: the role-split kernels (S sweep and G sweep on different waves, accumulators and weights handed over
    through LDS) run the same chains as the one-wave kernel (VQ_CE_NO_ROLES=1) -- equal bits, at ragged row counts, codebook
    tails and zero distances."""
    import os
    native = _native()
    x, cb = make_x((H, M, D), "S"), make_codebook(H, K, D, "S")
    if metric == 1:
        x = x * 0.25
    g = torch.Generator().manual_seed(5)
    target = torch.randint(0, K, (H, M), generator=g)
    target[:, ::7] = -1
    if metric == 0 and M > 40:
        x[0, 40] = cb[0, K // 2]  # a zero distance
    xs, cbs, ts = x.cuda(), cb.cuda(), target.cuda()
    lse, tl = native.softmax_stats(xs, cbs, metric=metric, target=ts)
    coef = torch.tensor([0.5], device="cuda")
    got = native.ce_backward(xs, cbs, lse, tl, ts, coef, metric=metric)
    os.environ["VQ_CE_NO_ROLES"] = "1"
    try:
        ref = native.ce_backward(xs, cbs, lse, tl, ts, coef, metric=metric)
    finally:
        os.environ.pop("VQ_CE_NO_ROLES", None)
    torch.cuda.synchronize()
    assert torch.equal(got.view(torch.int32), ref.view(torch.int32))


def test_fused_cross_entropy_backward_strided_heads():
    native = _native()
    rows, heads, d, K = 200, 3, 32, 96
    x4 = make_x((rows, heads, d), "S")
    cb = make_codebook(heads, K, d, "S")
    g = torch.Generator().manual_seed(2)
    target = torch.randint(0, K, (rows, heads), generator=g)
    flat = x4.cuda().permute(1, 0, 2)
    ts = target.cuda().permute(1, 0)  # strided [h, rows]
    lse, tl = native.softmax_stats(flat, cb.cuda(), target=ts)
    got = native.ce_backward(flat, cb.cuda(), lse, tl, ts, torch.tensor([1.0 / rows], device="cuda"))
    want = _ce_grad_reference(x4.permute(1, 0, 2).contiguous(), cb, target.permute(1, 0).contiguous(), 0, 1.0 / rows)
    torch.cuda.synchronize()
    np.testing.assert_allclose(got.cpu().double().numpy(), want.numpy(), atol=2e-5 * float(want.abs().max()), rtol=2e-4)


@pytest.mark.parametrize("H,M,K,D", [s for s in SHAPES if s[3] <= 512] + [(1, 8192, 256, 64), (8, 256, 8192, 64)])
@pytest.mark.parametrize("metric", [0, 1])
def test_search_with_fused_log_sum_exp(oracle, H, M, K, D, metric):
    """vq_quantize_lse_f32: same idx / best / out bits as vq_quantize_f32, plus the row's log-sum-exp (float64 check)."""
    native = _native()
    x, cb = make_x((H, M, D), "S"), make_codebook(H, K, D, "S")
    if metric == 1:
        x = x * 0.25
    xs, cbs = x.cuda(), cb.cuda()[:, None].contiguous()
    plain = native.quantize(xs, cbs, metric=metric, ste=True, want_sq_err=True)
    fused = native.quantize(xs, cbs, metric=metric, ste=True, want_sq_err=True, want_lse=True)
    torch.cuda.synchronize()
    for key in ("idx", "best", "out"):
        assert torch.equal(plain[key], fused[key]), key
    # the plain call may take the split-K path (few rows): same terms, different summation order
    torch.testing.assert_close(plain["sq_err"], fused["sq_err"], rtol=1e-6, atol=0)
    sims = _oracle_sims(oracle, x, cb, metric).astype(np.float64)
    mx = sims.max(-1, keepdims=True)
    lse = (mx + np.log(np.exp(sims - mx).sum(-1, keepdims=True)))[..., 0]
    np.testing.assert_allclose(fused["lse"].cpu().numpy(), lse, rtol=2e-6, atol=2e-5)
