"""GPU: the resident-codebook kernel (csrc/vq_search_resident.inc) -- small codebooks whose packed image stays in LDS while the
waves stream 32-row blocks past it (BASELINE configs[0]'s class: K = 256, D = 64).  Same arithmetic as vq_search_mfma, so every
row must equal the one-thread-per-row kernel bit for bit, and an oracle sample pins both; ragged row counts, dims that
are not the padded width, one code, several heads as strided views, the dot metric, non-finite rows and codes."""
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


def _same(a, b):
    an, bn = torch.isnan(a), torch.isnan(b)
    return bool(torch.equal(an, bn)) and bool(torch.equal(torch.where(an, torch.zeros_like(a), a).view(torch.int32),
                                                            torch.where(bn, torch.zeros_like(b), b).view(torch.int32)))


SHAPES = [
    # (H, M, K, D)
    (1, 140000, 256, 64),      # cfg1's codebook at a row count that fills the chip
    (1, 262144, 256, 64),
    (1, 150001, 256, 128),     # Dp = 128: one accumulator, image 135 KB
    (1, 140000, 224, 64),      # K not a multiple of 32 * 8
    (1, 140077, 256, 32),
    (1, 140000, 100, 48),      # D is not the padded width (Dp = 64, one padding slab), K not a multiple of 32
    (1, 140000, 1, 16),        # one code
    (1, 131072, 33, 128),
    (4, 40000, 256, 64),       # heads (grid.y), 160 000 rows in all
    (1, 1000000, 64, 32),      # many blocks per wave, short sweeps (the side jobs finish in the open)
]


@pytest.mark.parametrize("H,M,K,D", SHAPES)
@pytest.mark.parametrize("metric", [0, 1])
def test_resident_kernel_equals_scalar_kernel_and_oracle(oracle, H, M, K, D, metric):
    native = _native()
    g = torch.Generator().manual_seed(H + M + K + D)
    x = torch.randn((H, M, D), generator=g).to(DEV)
    cb = torch.randn((H, 1, K, D), generator=g).to(DEV)
    r = native.quantize(x, cb, metric=metric, want_best=True)
    s = native.quantize(x, cb, metric=metric, want_best=True, flags=native.F_FORCE_SIMPLE)
    torch.cuda.synchronize()
    assert torch.equal(r["idx"], s["idx"])
    assert torch.equal(r["best"].view(torch.int32), s["best"].view(torch.int32))
    assert torch.equal(r["out"], s["out"])
    hh = torch.arange(H, device=DEV)[:, None]
    assert torch.equal(r["out"], cb[:, 0][hh, r["idx"][..., 0]])
    rows = torch.cat([torch.arange(0, 300), torch.arange(M - 300, M), torch.randint(0, M, (400,), generator=g)])
    for h in range(H):
        ri, rb = oracle.nearest(x[h, rows].cpu().numpy(), cb[h, 0].cpu().numpy(), metric)
        np.testing.assert_array_equal(r["idx"][h, rows, 0].cpu().numpy(), ri)
        assert np.array_equal(r["best"][h, rows, 0].cpu().numpy().view(np.uint32), rb.view(np.uint32))


def test_resident_kernel_on_strided_head_views_through_the_module():
    """VectorQuantize with heads hands [heads, rows, dim] VIEWS of one buffer (row stride = heads * dim) and [rows, heads]
    ordered output buffers: the slab DMA and the hidden copy must honour the strides."""
    import vector_quantization as vq
    from vector_quantization.codebooks import CodebookParams

    torch.manual_seed(3)
    mod = vq.VectorQuantize(dim=128, heads=2, codebook_dim=64, separate_codebook_per_head=True,
                            codebook_params=CodebookParams(dim=64, codebook_size=256)).to(DEV).eval()
    x = torch.randn((80, 1024, 128), generator=torch.Generator().manual_seed(4)).to(DEV)
    with torch.no_grad():
        q, idx, _ = mod(x)
    cbs = mod._codebook.embeddings
    want = torch.cat([cbs[h][idx[..., h]] for h in range(2)], dim=-1)
    assert torch.equal(q, want)
    native = _native()
    flat = x.reshape(-1, 2, 64).permute(1, 0, 2)
    s = native.quantize(flat, cbs.detach()[:, None].contiguous(), flags=native.F_FORCE_SIMPLE)
    assert torch.equal(idx.reshape(-1, 2).t(), s["idx"][..., 0])


@pytest.mark.parametrize("what", ["rows", "codes"])
def test_resident_kernel_non_finite_inputs(oracle, what):
    native = _native()
    M, K, D = 140000, 256, 64
    g = torch.Generator().manual_seed(11)
    x = torch.randn((M, D), generator=g)
    cb = torch.randn((K, D), generator=g)
    if what == "rows":
        for r_, d_, v in ((0, 3, float("nan")), (31, 63, float("inf")), (32, 0, float("-inf")), (70000, 5, float("nan")),
                          (M - 1, 9, float("inf")), (M - 33, 1, float("nan"))):
            x[r_, d_] = v
    else:
        cb[200, 7] = float("nan")
        cb[90, 0] = float("nan")
    xd, cbd = x.to(DEV)[None], cb.to(DEV)[None, None].contiguous()
    r = native.quantize(xd, cbd, want_best=True)
    s = native.quantize(xd, cbd, want_best=True, flags=native.F_FORCE_SIMPLE)
    torch.cuda.synchronize()
    assert torch.equal(r["idx"], s["idx"]) and _same(r["best"], s["best"]) and _same(r["out"], s["out"])
    rows = [0, 31, 32, 70000, M - 1, M - 33] + list(range(100, 400))
    ri, rb = oracle.nearest(x[rows].numpy(), cb.numpy(), oracle.EUCLID)
    np.testing.assert_array_equal(r["idx"][0, rows, 0].cpu().numpy(), ri)
    if what == "codes":
        assert bool((r["idx"] == 90).all())


def test_the_resident_kernel_is_the_one_that_runs():
    from torch.profiler import ProfilerActivity, profile

    native = _native()
    g = torch.Generator().manual_seed(1)
    x = torch.randn((1, 262144, 64), generator=g).to(DEV)
    cb = torch.randn((1, 1, 256, 64), generator=g).to(DEV)
    packed = native.pack_codebooks(cb, 0)

    def names(xx, **kw):
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
            native.quantize(xx, cb, packed=packed, **kw)
            torch.cuda.synchronize()
        return [ev.name for ev in prof.events() if str(getattr(ev, "device_type", "")).endswith("CUDA") and "vq_" in ev.name]

    plain = names(x)
    if not plain:
        pytest.skip("torch.profiler reported no device activity on this build")
    assert any("vq_search_resident" in n for n in plain), plain
    train = names(x, ste=True, want_sq_err=True)
    assert not any("vq_search_resident" in n for n in train), train
    small = names(x[:, :8192])  # cfg1's own row count: too few rows to give every wave slot a block
    assert not any("vq_search_resident" in n for n in small), small
