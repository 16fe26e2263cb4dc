"""GPU: the launch plans for row counts that do not fill whole rounds of workgroups (csrc/vq_kernels.hip plan_k_split /
plan_main_tail): K split over several workgroups per row block, or whole rounds fused + the remainder as its own K-split call.
Results must not depend on the plan: indices, winning distances and outputs equal the one-thread-per-row kernel on every row,
the squared-error sum (one sum over both parts of a two-call plan) agrees to 1e-6, eval and training outputs alike."""
from __future__ import annotations

import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"

CASES = [
    # (H, M, K, D, metric)          what the planner does with it on a 256-CU device
    (1, 70000, 1024, 256, 0),      # 274 row blocks: 256 fused + 18 as a K-split tail
    (1, 131073, 1024, 256, 0),     # 513 row blocks, the last one holds ONE row
    (1, 66000, 8192, 256, 0),      # 258 row blocks, long sweep
    (1, 40000, 4096, 256, 1),      # 157 row blocks (< one round): K split 3 ways
    (1, 40000, 4096, 512, 0),      # wave-pair kernel in keys mode
    (1, 70000, 8192, 64, 0),       # Dp = 64
    (1, 200000, 1024, 64, 1),
    (1, 150000, 4096, 128, 0),
    (8, 10000, 8192, 64, 0),       # several heads: 320 workgroups = 256 fused (32 row blocks of every head) + a tail
    (4, 20000, 2048, 256, 1),
    (6, 9000, 1024, 512, 0),       # heads that do not divide the CU count
]


@pytest.mark.parametrize("H,M,K,D,metric", CASES)
@pytest.mark.parametrize("training", [False, True])
def test_planned_launches_equal_scalar_kernel(H, M, K, D, metric, training):
    from vector_quantization import native

    native.load()
    g = torch.Generator(device=DEV).manual_seed(M + K + D)
    x = torch.randn((H, M, D), device=DEV, generator=g)
    cb = torch.randn((H, 1, K, D), device=DEV, generator=g)
    r = native.quantize(x, cb, metric=metric, ste=training, want_sq_err=training)
    s = native.quantize(x, cb, metric=metric, ste=training, want_sq_err=training, flags=native.F_FORCE_SIMPLE)
    torch.cuda.synchronize()
    assert torch.equal(r["idx"], s["idx"])
    assert torch.equal(r["best"].view(torch.int32), s["best"].view(torch.int32))
    assert torch.equal(r["out"], s["out"])
    if training:
        torch.testing.assert_close(r["sq_err"], s["sq_err"], rtol=1e-6, atol=0)


def test_module_forward_with_planned_launches_matches_scalar_search():
    """The modules hand strided views (channel-first input, [rows, heads]-ordered index buffers) to the planned launches."""
    import vector_quantization as vq
    from vector_quantization import native
    from vector_quantization.codebooks import CodebookParams

    torch.manual_seed(0)
    for channel_last in (True, False):
        mod = vq.VectorQuantize(dim=256, channel_last=channel_last,
                                codebook_params=CodebookParams(dim=256, codebook_size=1024)).to(DEV).eval()
        x = torch.randn(70, 1000, 256, device=DEV)  # 70 000 rows: whole rounds fused + a K-split tail
        xin = x if channel_last else x.permute(0, 2, 1).contiguous()
        with torch.no_grad():
            q, idx, _ = mod(xin)
        flat = x.reshape(1, -1, 256)
        ref = native.quantize(flat, mod._codebook.embeddings.detach()[:, None].contiguous(), flags=native.F_FORCE_SIMPLE)
        assert torch.equal(idx.reshape(-1), ref["idx"].reshape(-1))
        qq = q if channel_last else q.permute(0, 2, 1)
        assert torch.equal(qq.reshape(-1, 256), ref["out"].reshape(-1, 256))


@pytest.mark.parametrize("H,M,K,D,metric", [(1, 300, 4096, 128, 0), (1, 8192, 8192, 512, 0), (2, 500, 2048, 64, 1),
                                            (1, 65536, 1024, 256, 0), (1, 200, 300, 700, 0)])
def test_key_planes_need_no_init_and_no_atomics(H, M, K, D, metric):
    """vq_search_key_planes_f32: a K split stores plane by plane (P = vq_key_planes > 1 when few rows would leave the chip
    empty); the MIN over the planes -- taken by vq_finalize_key_planes_f32 on the fly -- is the one-plane atomic result, and
    planes of several shards simply stack."""
    from vector_quantization import native

    native.load()
    g = torch.Generator(device=DEV).manual_seed(M + K)
    x = torch.randn((H, M, D), device=DEV, generator=g)
    cb = torch.randn((H, K, D), device=DEV, generator=g)
    ref = torch.empty((H, M), dtype=torch.int64, device=DEV)
    native.keys_init(ref)
    native.search_keys(x, cb, ref, metric=metric)
    planes = native.search_key_planes(x, cb, metric=metric)
    assert planes.dim() == 3 and tuple(planes.shape[1:]) == (H, M)
    if M <= 8192 and D <= 512:
        assert planes.shape[0] > 1, "few rows: K must have been split over workgroups"
    assert torch.equal(planes.amin(dim=0), ref)
    fin1 = native.finalize_keys(x, cb, ref, metric=metric)
    finp = native.finalize_keys(x, cb, planes, metric=metric)
    for k in ("out", "idx", "best"):
        assert torch.equal(fin1[k], finp[k]), k
    # two shards, planes stacked: the finalize's MIN over 2 P planes is the full search
    half = K // 2
    p0 = native.search_key_planes(x, cb[:, :half].contiguous(), metric=metric, idx_offset=0)
    p1 = native.search_key_planes(x, cb[:, half:].contiguous(), metric=metric, idx_offset=half)
    fin2 = native.finalize_keys(x, cb, torch.cat([p0, p1], dim=0), metric=metric)
    assert torch.equal(fin2["idx"], fin1["idx"]) and torch.equal(fin2["out"], fin1["out"])


@pytest.mark.parametrize("M,K,D,Q,H", [(70000, 1024, 256, 4, 1), (80000, 256, 256, 3, 1), (140000, 512, 200, 2, 1), (65536, 1024, 256, 3, 1),
                                       (66000, 2048, 256, 3, 1), (36000, 1024, 128, 3, 2), (34000, 1024, 400, 2, 1), (35001, 4096, 64, 2, 2)])
@pytest.mark.parametrize("training", [False, True])
def test_residual_stacks_with_awkward_row_counts(M, K, D, Q, H, training):
    """Residual stacks cannot split K inside one launch; a row count just above a multiple of 256 x CUs runs its whole rounds on the
    fused kernel and the remainder STAGE BY STAGE (K split over all CUs, the finalize writing the next residual), or -- short
    sweeps -- on 128-row workgroups.  Whatever the plan: equal to the stages searched one by one with the one-thread-per-row
    kernel and the reference's residual arithmetic (residual_vq.py:232-233, vector_quantize_pytorch.py:273)."""
    from vector_quantization import native

    native.load()
    g = torch.Generator(device=DEV).manual_seed(M + K + Q)
    x = torch.randn((H, M, D), device=DEV, generator=g)
    cbs = torch.stack([torch.stack([torch.randn((K, D), device=DEV, generator=g) * 2.0 ** (-i / 2.0) for i in range(Q)])
                       for _ in range(H)]).contiguous()
    r = native.quantize(x, cbs, ste=training, want_sq_err=training, want_best=True)
    res, out = x, torch.zeros_like(x)
    for q in range(Q):
        s = native.quantize(res, cbs[:, q:q + 1].contiguous(), flags=native.F_FORCE_SIMPLE, want_best=True)
        assert torch.equal(r["idx"][..., q], s["idx"][..., 0]), f"stage {q}"
        assert torch.equal(r["best"][..., q].view(torch.int32), s["best"][..., 0].view(torch.int32)), f"stage {q}"
        code = s["out"]
        quant = res + (code - res) if training else code
        if training:
            err = (code - res).double().pow(2).sum()
            torch.testing.assert_close(r["sq_err"][q], err, rtol=1e-6, atol=0)
        res = res - quant
        out = out + quant
    assert torch.equal(r["out"], out)


@pytest.mark.parametrize("M,K,D,Q,H", [(3000, 2048, 64, 3, 3), (36000, 1024, 128, 2, 2), (70000, 1024, 256, 2, 1)])
def test_residual_stacks_per_head_squared_errors_under_both_plans(M, K, D, Q, H, residual_plan):
    """GroupedResidualVQ's per-group losses (VQ_F_SQERR_PER_HEAD, sq_err [H][Q]): the stage-by-stage plan adds each stage's
    per-head sums where the fused launch writes them -- same indices and outputs, sums equal to 1e-6."""
    from vector_quantization import native

    native.load()
    g = torch.Generator(device=DEV).manual_seed(M + K + Q + H)
    x = torch.randn((H, M, D), device=DEV, generator=g)
    cbs = torch.stack([torch.stack([torch.randn((K, D), device=DEV, generator=g) * 2.0 ** (-i / 2.0) for i in range(Q)])
                       for _ in range(H)]).contiguous()
    r = native.quantize(x, cbs, ste=True, want_sq_err=True, sq_err_per_head=True)
    res = x
    for q in range(Q):
        hh = torch.arange(H, device=DEV)[:, None]
        code = cbs[:, q][hh, r["idx"][..., q]]
        err = (code - res).double().pow(2).sum(dim=(1, 2))
        torch.testing.assert_close(r["sq_err"][:, q], err, rtol=1e-6, atol=0)
        res = res - (res + (code - res))
    s = native.quantize(x, cbs, ste=True, want_sq_err=True)
    assert torch.equal(r["idx"], s["idx"]) and torch.equal(r["out"], s["out"])
    torch.testing.assert_close(r["sq_err"].sum(0), s["sq_err"], rtol=1e-6, atol=0)
