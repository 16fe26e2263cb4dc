"""GPU: the persistent inference kernel (vq_search_persist: Dp = 256, plain eval call, 32..96 sub-tiles per sweep, at least
two row blocks per CU), which copies block b's winners during block b + 1's sweep.  Every row is compared with the
one-block-per-workgroup kernel (the same call with want_sq_err=True takes that path) and with the scalar kernel; a row
sample with the CPU oracle.  Ragged sizes: partial last block, blocks not divisible by the grid, K % 32 != 0, D < 256,
two heads sharing the CUs, strided multi-head views, both metrics."""
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


@pytest.mark.parametrize("H,M,K,D,metric", [
    (1, 262144, 1024, 256, 0),
    (1, 140001, 1000, 256, 0),
    (1, 200000, 2048, 200, 0),
    (2, 70000, 1100, 256, 1),
    (1, 135000, 3000, 132, 0),
])
def test_persistent_kernel_equals_one_block_kernel(oracle, H, M, K, D, metric):
    native = _native()
    g = torch.Generator().manual_seed(M + K)
    x = torch.randn((H, M, D), generator=g).to(DEV)
    cb = torch.randn((H, 1, K, D), generator=g).to(DEV)
    import os
    a = native.quantize(x, cb, metric=metric)                               # persistent kernel
    t = native.quantize(x, cb, metric=metric, ste=True, want_sq_err=True)   # persistent kernel, training-mode copy (round 3)
    l = native.quantize(x, cb, metric=metric, want_sq_err=True)             # ... squared error only
    os.environ["VQ_NO_PERSIST_TRAIN"] = "1"
    try:
        b = native.quantize(x, cb, metric=metric, want_sq_err=True)             # one block per workgroup
        bt = native.quantize(x, cb, metric=metric, ste=True, want_sq_err=True)  # ... straight-through + squared error
    finally:
        os.environ.pop("VQ_NO_PERSIST_TRAIN", None)
    s = native.quantize(x, cb, metric=metric, flags=native.F_FORCE_SIMPLE)  # scalar kernel
    torch.cuda.synchronize()
    for other in (b, s, l):
        assert torch.equal(a["idx"], other["idx"])
        assert torch.equal(a["best"].view(torch.int32), other["best"].view(torch.int32))
        assert torch.equal(a["out"], other["out"])
    assert torch.equal(t["idx"], bt["idx"]) and torch.equal(t["out"], bt["out"])  # x + (c - x): the same bits on both kernels
    assert torch.equal(t["best"].view(torch.int32), bt["best"].view(torch.int32))
    torch.testing.assert_close(t["sq_err"], bt["sq_err"], rtol=1e-6, atol=0)
    torch.testing.assert_close(l["sq_err"], b["sq_err"], rtol=1e-6, atol=0)
    want_err = (cb[:, 0][torch.arange(H, device=DEV)[:, None], a["idx"][..., 0]] - x).double().pow(2).sum()
    torch.testing.assert_close(t["sq_err"].sum(), want_err, rtol=1e-6, atol=0)
    hh = torch.arange(H, device=DEV)[:, None]
    assert torch.equal(a["out"], cb[:, 0][hh, a["idx"][..., 0]])
    rows = torch.cat([torch.randperm(M, generator=torch.Generator().manual_seed(2))[:300], torch.arange(M - 50, M)])
    for h in range(H):
        ri, rb = oracle.nearest(x[h, rows].cpu().numpy(), cb[h, 0].cpu().numpy(), metric)
        np.testing.assert_array_equal(a["idx"][h, rows, 0].cpu().numpy(), ri)
        assert np.array_equal(a["best"][h, rows, 0].cpu().numpy().view(np.uint32), rb.view(np.uint32))


def test_persistent_kernel_on_strided_head_views(oracle):
    """The module's layout: x [rows, heads * d] searched in place as [heads, rows, d] views; out / idx strided as well."""
    import vector_quantization as vq
    from vector_quantization.codebooks import CodebookParams

    torch.manual_seed(0)
    mod = vq.VectorQuantize(dim=512, codebook_params=CodebookParams(dim=256, codebook_size=1024), heads=2, codebook_dim=256,
                            separate_codebook_per_head=True).to(DEV).eval()
    x = torch.randn((70, 1024, 512), generator=torch.Generator().manual_seed(4)).to(DEV)
    with torch.no_grad():
        q, idx, _ = mod(x)
        mod.train()
        q2, idx2, _ = mod(x, freeze_codebook=True)   # training forward: straight-through + loss (the persistent kernel's TRAIN variant)
    assert torch.equal(idx, idx2)
    torch.testing.assert_close(q, q2, rtol=0, atol=1e-6)  # x + (c - x) vs c
    cbs = mod._codebook.embeddings
    want = torch.cat([cbs[h][idx[..., h]] for h in range(2)], dim=-1)
    assert torch.equal(q, want)


def test_the_persistent_kernel_is_the_one_that_runs():
    """Guards the launcher's selection: the headline shape (cfg2) must go through vq_search_persist, eval and training call alike."""
    from torch.profiler import ProfilerActivity, profile

    native = _native()
    g = torch.Generator().manual_seed(1)
    x = torch.randn((1, 262144, 256), generator=g).to(DEV)
    cb = torch.randn((1, 1, 1024, 256), generator=g).to(DEV)
    packed = native.pack_codebooks(cb, 0)

    def names(**kw):
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
            native.quantize(x, cb, packed=packed, **kw)
            torch.cuda.synchronize()
        return [ev.name for ev in prof.events() if str(getattr(ev, "device_type", "")).endswith("CUDA") and "vq_" in ev.name]

    plain = names()
    if not plain:
        pytest.skip("torch.profiler reported no device activity on this build")
    assert any("vq_search_persist" in n for n in plain), plain
    train = names(ste=True, want_sq_err=True)  # round 3: the training-mode call runs the persistent kernel's TRAIN variant
    assert any("vq_search_persist" in n for n in train), train
    import os
    os.environ["VQ_NO_PERSIST_TRAIN"] = "1"
    try:
        train1 = names(ste=True, want_sq_err=True)
    finally:
        os.environ.pop("VQ_NO_PERSIST_TRAIN", None)
    assert not any("vq_search_persist" in n for n in train1) and any("vq_search_mfma" in n for n in train1), train1
