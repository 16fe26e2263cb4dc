"""GPU: the atomics-free EMA accumulation (vq_ema_accumulate_det_f32) -- counts exact, sums equal to an fp64 scatter-add to
fp32 rounding, and BIT-IDENTICAL from run to run (the default path adds with float atomics, whose order varies);
torch.use_deterministic_algorithms(True) selects it inside the modules (codebooks.py:405-415 is the step it replaces)."""
from __future__ import annotations

import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _native():
    from vector_quantization import native

    native.load()
    return native


CASES = [
    # (H, M, K, D, masked)
    (1, 65536, 1024, 256, False),
    (4, 20000, 300, 64, False),
    (1, 5000, 40, 100, True),      # D % 4 != 0, masked rows
    (1, 100, 16, 32, False),       # one row block
    (1, 2000, 8192, 64, False),    # many codes, few rows (the atomic path's regime: still correct here)
    (2, 30000, 77, 600, True),     # D > 256: two passes per row
]


@pytest.mark.parametrize("H,M,K,D,masked", CASES)
def test_det_accumulate_exact_and_reproducible(H, M, K, D, masked):
    native = _native()
    g = torch.Generator().manual_seed(M + K + D)
    x = torch.randn((H, M, D), generator=g).to(DEV)
    idx = torch.randint(0, K, (H, M), generator=g).to(DEV)
    mask = (torch.rand((H, M), generator=g) < 0.7).to(DEV) if masked else None
    runs = [native.ema_accumulate(x, idx, K, mask, deterministic=True) for _ in range(4)]
    torch.cuda.synchronize()
    for c, s in runs[1:]:
        assert torch.equal(c, runs[0][0]) and torch.equal(s.view(torch.int32), runs[0][1].view(torch.int32))
    counts, sums = runs[0]
    w = torch.ones((H, M), device=DEV, dtype=torch.float64) if mask is None else mask.double()
    ref_c = torch.zeros((H, K), dtype=torch.float64, device=DEV).scatter_add_(1, idx, w)
    ref_s = torch.zeros((H, K, D), dtype=torch.float64, device=DEV).scatter_add_(
        1, idx[..., None].expand(-1, -1, D), x.double() * w[..., None])
    assert torch.equal(counts.double(), ref_c)
    scale = ref_s.abs().amax().clamp(min=1.0)
    assert float((sums.double() - ref_s).abs().amax() / scale) < 2e-6
    # the default (atomic) path agrees to summation-order rounding
    c2, s2 = native.ema_accumulate(x, idx, K, mask)
    assert torch.equal(c2, counts)
    torch.testing.assert_close(s2, sums, rtol=1e-4, atol=1e-4 * float(scale))


def test_det_accumulate_strided_views():
    native = _native()
    g = torch.Generator().manual_seed(1)
    H, M, K, D = 2, 9000, 50, 48
    buf = torch.randn((M, H, D + 8), generator=g).to(DEV)
    x = buf[..., :D].permute(1, 0, 2)                      # [H, M, D] with row stride H * (D + 8)
    idx_all = torch.randint(0, K, (M, H, 3), generator=g).to(DEV)
    idx = idx_all[..., 1].permute(1, 0)                    # strided indices
    a = native.ema_accumulate(x, idx, K, None, deterministic=True)
    b = native.ema_accumulate(x.contiguous(), idx.contiguous(), K, None, deterministic=True)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])


def _train_twice(make, x, steps=3):
    states = []
    for _ in range(2):
        torch.manual_seed(0)
        mod = make().to(DEV).train()
        for _ in range(steps):
            mod(x)
        torch.cuda.synchronize()
        states.append({k: v.clone() for k, v in mod.state_dict().items()})
    return states


def test_modules_train_reproducibly_under_torch_deterministic_mode():
    import vector_quantization as vq
    from vector_quantization.codebooks import CodebookParams

    torch.manual_seed(1)
    x = torch.randn(16, 1024, 64, device=DEV)
    makers = (
        lambda: vq.VectorQuantize(dim=64, codebook_params=CodebookParams(dim=64, codebook_size=256, threshold_ema_dead_code=0)),
        lambda: vq.ResidualVQ(dim=64, num_quantizers=3,
                              codebook_params=CodebookParams(dim=64, codebook_size=128, threshold_ema_dead_code=0)),
    )
    prev = torch.are_deterministic_algorithms_enabled()
    torch.use_deterministic_algorithms(True)
    try:
        for make in makers:
            a, b = _train_twice(make, x)
            for k in a:
                assert torch.equal(a[k], b[k]), f"{k} differs between two identical runs"
    finally:
        torch.use_deterministic_algorithms(prev)


def test_det_accumulate_rejects_small_workspace():
    import ctypes

    native = _native()
    lib = native.load()
    H, M, K, D = 1, 4096, 64, 32
    need = lib.vq_ema_det_workspace_bytes(H, M, K, D)
    assert need > 0 and lib.vq_ema_det_workspace_bytes(H, M, K, 4096) == 0
    x = torch.randn((H, M, D), device=DEV)
    idx = torch.zeros((H, M), dtype=torch.int64, device=DEV)
    counts = torch.zeros((H, K), device=DEV)
    sums = torch.zeros((H, K, D), device=DEV)
    ws = torch.empty(max(need // 2, 16), dtype=torch.uint8, device=DEV)
    rc = lib.vq_ema_accumulate_det_f32(x.data_ptr(), D, M * D, idx.data_ptr(), 1, M, None, H, M, K, D, counts.data_ptr(),
                                       sums.data_ptr(), ws.data_ptr(), ws.numel(), None)
    assert rc == -1 and b"vq_ema_det_workspace_bytes" in lib.vq_last_error()
    torch.cuda.synchronize()
