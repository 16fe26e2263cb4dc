"""GPU: BASELINE full sizes through size-independent properties, plus edge cases.

At M = 262 144 the CPU oracle would take minutes, so full-size runs are checked by (a) oracle comparison on a
random sample of rows (rows are independent), (b) idempotence: quantising the quantised rows returns them with
distance exactly 0, (c) path equivalence: fused == split-K == scalar kernel == two-shard key merge,
(d) the distance identity |x - q| == best, (e) residual round trip out + r_Q == x.
"""
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


def _rand(shape, seed):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed))


@pytest.mark.parametrize("M,K,D,H", [(262144, 1024, 256, 1), (262144, 8192, 256, 1), (65536, 8192, 64, 8),
                                     (65536, 8192, 512, 8)])  # cfg2, north-star shape, cfg3a, cfg3b at BASELINE size
def test_full_size_sample_idempotence_identity(oracle, M, K, D, H):
    native = _native()
    x = _rand((H, M, D), 1234).to(DEV)
    cb = _rand((H, 1, K, D), 4321).to(DEV)
    r = native.quantize(x, cb, want_sq_err=True)
    idx, best, out = r["idx"][..., 0], r["best"][..., 0], r["out"]
    # (a) oracle on a sample of rows
    g = torch.Generator().manual_seed(7)
    rows = torch.randperm(M, generator=g)[:(2048 if D <= 256 else 768)]
    if D > 256:  # cfg3b (fused multi-head wave-pair launch): also every row of a slice against the one-thread-per-row kernel
        sl = native.quantize(x[:, 1000:3048], cb, flags=native.F_FORCE_SIMPLE)
        assert torch.equal(sl["idx"][..., 0], idx[:, 1000:3048])
        assert torch.equal(sl["best"][..., 0].view(torch.int32), best[:, 1000:3048].view(torch.int32))
    for h in range(H):
        ri, rb = oracle.nearest(x[h, rows].cpu().numpy(), cb[h, 0].cpu().numpy(), oracle.EUCLID)
        np.testing.assert_array_equal(idx[h, rows].cpu().numpy(), ri)
        assert np.array_equal(best[h, rows].cpu().numpy().view(np.uint32), rb.view(np.uint32))
    # exact gather
    hh = torch.arange(H, device=DEV)[:, None]
    assert torch.equal(out, cb[:, 0][hh, idx])
    # (d) distance identity and squared-error sum
    d = (x - out).double().pow(2).sum(-1)
    torch.testing.assert_close(d.sqrt().float(), best, rtol=2e-3, atol=2e-3)
    torch.testing.assert_close(r["sq_err"][0], d.sum(), rtol=1e-6, atol=0)
    # (b) idempotence: codes quantise to themselves at distance exactly 0
    r2 = native.quantize(out, cb)
    assert torch.equal(r2["out"], out)
    assert float(r2["best"].abs().max()) == 0.0
    assert bool((r2["idx"][..., 0] <= idx).all())


@pytest.mark.parametrize("M,K,D", [(262144, 1024, 256), (8192, 65536, 512), (5000, 777, 100)])
def test_path_equivalence_fused_split_scalar_sharded(M, K, D):
    native = _native()
    x = _rand((1, M, D), 11).to(DEV)
    cb = _rand((1, 1, K, D), 12).to(DEV)
    base = native.quantize(x, cb)
    split = native.quantize(x, cb, flags=native.F_FORCE_SPLIT)
    assert torch.equal(base["idx"], split["idx"]) and torch.equal(base["best"], split["best"])
    assert torch.equal(base["out"], split["out"])
    if M * K <= 5000 * 1024:
        simple = native.quantize(x, cb, flags=native.F_FORCE_SIMPLE)
        assert torch.equal(base["idx"], simple["idx"]) and torch.equal(base["best"], simple["best"])
    # two-shard key merge on one GPU == RCCL MIN all-reduce of the same keys
    keys = torch.empty((1, M), dtype=torch.int64, device=DEV)
    native.keys_init(keys)
    k0 = (K // 2 + 31) // 32 * 32 if K > 64 else K // 2
    native.search_keys(x, cb[:, 0, :k0].contiguous(), keys, idx_offset=0)
    keys2 = torch.empty_like(keys)
    native.keys_init(keys2)
    native.search_keys(x, cb[:, 0, k0:].contiguous(), keys2, idx_offset=k0)
    merged = torch.minimum(keys, keys2)
    fin = native.finalize_keys(x, cb[:, 0].contiguous(), merged)
    assert torch.equal(fin["idx"], base["idx"][..., 0]) and torch.equal(fin["best"], base["best"][..., 0])
    assert torch.equal(fin["out"], base["out"])


def test_rows_are_independent_full_cfg4():
    """ResidualVQ cfg4 at full size: processing the batch in two halves gives the same bits; round trip holds."""
    native = _native()
    M, K, D, Q = 65536, 1024, 256, 8
    x = _rand((1, M, D), 3).to(DEV)
    cbs = torch.stack([_rand((K, D), 100 + i) * 2.0 ** (-i / 2.0) for i in range(Q)])[None].to(DEV)
    full = native.quantize(x, cbs, want_sq_err=True)
    a = native.quantize(x[:, : M // 2], cbs)
    b = native.quantize(x[:, M // 2:], cbs)
    assert torch.equal(full["idx"], torch.cat([a["idx"], b["idx"]], dim=1))
    assert torch.equal(full["out"], torch.cat([a["out"], b["out"]], dim=1))
    # residual chain recomputed with torch ops from the indices reproduces out exactly, and out + r_Q == x
    r = x[0]
    out = torch.zeros_like(r)
    errs = []
    for q in range(Q):
        c = cbs[0, q][full["idx"][0, :, q]]
        errs.append(float((c - r).double().pow(2).sum()))
        r = r - c
        out = out + c
    assert torch.equal(out, full["out"][0])
    torch.testing.assert_close(out + r, x[0], rtol=0, atol=1e-5)
    np.testing.assert_allclose(full["sq_err"].cpu().numpy(), errs, rtol=1e-6)


def test_empty_and_tiny_inputs(oracle):
    native = _native()
    cb = _rand((1, 1, 40, 24), 5).to(DEV)
    r = native.quantize(torch.empty((1, 0, 24), device=DEV), cb, want_sq_err=True)
    assert r["idx"].shape == (1, 0, 1) and r["out"].shape == (1, 0, 24) and float(r["sq_err"][0]) == 0.0
    for M in (1, 2, 31, 32, 33, 63, 64, 65, 255, 256, 257):
        x = _rand((1, M, 24), M)
        ref_i, ref_b = oracle.nearest(x[0].numpy(), cb[0, 0].cpu().numpy(), oracle.EUCLID)
        r = native.quantize(x.to(DEV), cb)
        np.testing.assert_array_equal(r["idx"][0, :, 0].cpu().numpy(), ref_i)
        assert np.array_equal(r["best"][0, :, 0].cpu().numpy().view(np.uint32), ref_b.view(np.uint32))


def test_modules_accept_empty_half_precision_and_noncontiguous(oracle):
    import vector_quantization as vq
    from vector_quantization.codebooks import CodebookParams

    torch.manual_seed(0)
    mod = vq.VectorQuantize(dim=32, codebook_params=CodebookParams(dim=32, codebook_size=64)).to(DEV).eval()
    q, i, loss = mod(torch.empty(0, 10, 32, device=DEV))
    assert q.shape == (0, 10, 32) and i.shape == (0, 10)
    x = torch.randn(4, 10, 64, device=DEV)[:, :, ::2]  # non-contiguous last dim
    q, i, _ = mod(x)
    ref_i, _ = oracle.nearest(x.reshape(-1, 32).cpu().numpy(), mod._codebook.embeddings[0].cpu().numpy(), oracle.EUCLID)
    np.testing.assert_array_equal(i.reshape(-1).cpu().numpy(), ref_i)
    xh = torch.randn(2, 7, 32, device=DEV).half()
    q, i, _ = mod(xh)  # the reference casts to fp32 (codebooks.py:354)
    ref_i, _ = oracle.nearest(xh.float().reshape(-1, 32).cpu().numpy(), mod._codebook.embeddings[0].cpu().numpy(),
                              oracle.EUCLID)
    np.testing.assert_array_equal(i.reshape(-1).cpu().numpy(), ref_i)
    assert q.dtype == torch.float32


def test_rows_wider_than_512_dims_take_the_sliced_sweep(oracle):
    native = _native()
    x = _rand((1, 100, 700), 1)
    cb = _rand((1, 1, 50, 700), 2)
    ref = oracle.vq_forward(x.numpy(), cb[:, 0].numpy(), oracle.EUCLID)
    r = native.quantize(x.to(DEV), cb.to(DEV), want_sq_err=True)
    np.testing.assert_array_equal(r["idx"][..., 0].cpu().numpy(), ref["idx"])
    assert np.array_equal(r["best"][..., 0].cpu().numpy().view(np.uint32), ref["best"].view(np.uint32))
    np.testing.assert_array_equal(r["out"].cpu().numpy(), ref["out"])


def test_sharded_search_single_rank_api(oracle):
    from vector_quantization.sharded import ShardedCodebookSearch

    full = _rand((4096, 128), 9).to(DEV)
    x = _rand((777, 128), 10).to(DEV)
    s = ShardedCodebookSearch(full)
    out, idx, best, sq = s(x, want_sq_err=True)
    ri, rb = oracle.nearest(x.cpu().numpy(), full.cpu().numpy(), oracle.EUCLID)
    np.testing.assert_array_equal(idx.cpu().numpy(), ri)
    assert np.array_equal(best.cpu().numpy().view(np.uint32), rb.view(np.uint32))
    assert torch.equal(out, full[idx])


def test_stream_reentrancy():
    """Launches follow the caller's current stream; two streams give the same bits as the default stream."""
    native = _native()
    x = _rand((1, 20000, 128), 1).to(DEV)
    cb = _rand((1, 1, 512, 128), 2).to(DEV)
    base = native.quantize(x, cb)
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    with torch.cuda.stream(s1):
        r1 = native.quantize(x, cb)
    with torch.cuda.stream(s2):
        r2 = native.quantize(x, cb)
    torch.cuda.synchronize()
    assert torch.equal(r1["idx"], base["idx"]) and torch.equal(r2["idx"], base["idx"])
    assert torch.equal(r1["out"], base["out"]) and torch.equal(r2["best"], base["best"])


def test_hip_graph_capture_and_replay(oracle):
    """The launch functions allocate nothing and never synchronise, so a module forward can be captured into a
    hipGraph (after one eager warm-up) and replayed on new data."""
    import vector_quantization as vq
    from vector_quantization.codebooks import CodebookParams

    torch.manual_seed(0)
    mod = vq.VectorQuantize(dim=64, codebook_params=CodebookParams(dim=64, codebook_size=256)).to(DEV).eval()
    static_x = torch.randn(8, 128, 64, device=DEV)
    with torch.no_grad():
        mod(static_x)  # warm-up: attributes, device info
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            q, i, _ = mod(static_x)
        for seed in (1, 2):
            xn = _rand((8, 128, 64), seed)
            static_x.copy_(xn.to(DEV))
            g.replay()
            torch.cuda.synchronize()
            ri, _ = oracle.nearest(xn.reshape(-1, 64).numpy(), mod._codebook.embeddings[0].cpu().numpy(), oracle.EUCLID)
            np.testing.assert_array_equal(i.reshape(-1).cpu().numpy(), ri)
            assert torch.equal(q.reshape(-1, 64), mod._codebook.embeddings[0][i.reshape(-1)])


@pytest.mark.parametrize("M,K,D", [(65536, 4096, 256), (32768, 2048, 64)])
def test_near_tie_heavy_default_init_large(oracle, M, K, D):
    """Reference-default (kaiming-uniform) codebooks with large-norm inputs: all K squared distances sit within a few
    hundred fp32 ulps of |x|^2, so rows are piles of exact and sqrt-only ties, many of them across sub-tiles.  The
    deferred tie resolution must still reproduce the oracle bit for bit."""
    from gen import make_codebook, make_x

    native = _native()
    x = make_x((1, M, D), "R", seed=77) * 64.0
    cb = make_codebook(1, K, D, "R", seed=78)
    ref_i, ref_b = oracle.nearest(x[0].numpy(), cb[0].numpy(), oracle.EUCLID)
    r = native.quantize(x.to(DEV), cb[:, None].contiguous().to(DEV))
    np.testing.assert_array_equal(r["idx"][0, :, 0].cpu().numpy(), ref_i)
    assert np.array_equal(r["best"][0, :, 0].cpu().numpy().view(np.uint32), ref_b.view(np.uint32))
    # how tie-heavy the case really is: rows whose two best sqrt distances are EQUAL in fp32 (sampled)
    sims = oracle.similarities(x[0, :512].numpy(), cb[0].numpy(), oracle.EUCLID)
    top2 = np.sort(-sims, axis=1)[:, :2]
    assert (top2[:, 0] == top2[:, 1]).mean() > 0.005


def test_integration_md_stub_runs_as_written(oracle):
    """The ctypes stub printed in INTEGRATION.md (what a maintainer of the reference would add) is executed verbatim."""
    import os
    import re

    native = _native()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    code = re.search(r"```python\n# vector_quantization/_mi355x.py.*?\n(.*?)```", text, flags=re.S).group(1)
    code = code.replace('ctypes.CDLL("libvq_mi355x.so")', f'ctypes.CDLL("{native.lib_path()}")')
    ns = {}
    exec(compile(code, "INTEGRATION.md", "exec"), ns)
    x = _rand((2, 700, 48), 21)
    cb = _rand((2, 130, 48), 22)
    out, idx = ns["nearest"](x.to(DEV), cb.to(DEV))
    torch.cuda.synchronize()
    ref = oracle.vq_forward(x.numpy(), cb.numpy(), oracle.EUCLID)
    np.testing.assert_array_equal(idx.cpu().numpy(), ref["idx"])
    np.testing.assert_array_equal(out.cpu().numpy(), ref["out"])


@pytest.mark.parametrize("H,M,K,D,masked", [
    (1, 262144, 1024, 256, False),  # cfg2: owner-computes kernel (a wave owns 8 codes)
    (1, 40000, 300, 100, True),     # owner kernel: 20 codes per wave, mask
    (3, 30000, 200, 64, False),     # owner kernel: heads (strided rows), 32 codes per wave
    (2, 20000, 100, 24, True),      # owner kernel: D smaller than a wave pass, unaligned row stride
    (1, 50000, 40, 600, False),     # owner kernel: D > 256 (two passes per row), 3 codes per wave
    (1, 65536, 1024, 256, False),   # too few rows per owner: memory-side atomics
    (1, 6000, 8192, 64, False),     # memory-side atomics
    (1, 1000, 256, 64, False),      # memory-side atomics
])
def test_ema_accumulate_paths_match_index_add(H, M, K, D, masked):
    """vq_ema_accumulate_f32 (owner-computes kernel and memory-side-atomic kernel) == index_add_ in float64."""
    from vector_quantization import native

    g = torch.Generator().manual_seed(H * 1000 + K)
    x4 = torch.randn((M, H, D), generator=g)
    x = x4.cuda().permute(1, 0, 2)                      # [H, M, D] with strided rows, like the head-split module view
    idx = torch.randint(0, K, (H, M), generator=g).cuda()
    mask = (torch.rand((H, M), generator=g) > 0.3).cuda() if masked else None
    counts, sums = native.ema_accumulate(x, idx, K, mask)
    torch.cuda.synchronize()
    w = mask.double() if masked else torch.ones((H, M), dtype=torch.float64, device="cuda")
    want_c = torch.zeros((H, K), dtype=torch.float64, device="cuda")
    want_s = torch.zeros((H, K, D), dtype=torch.float64, device="cuda")
    for h in range(H):
        want_c[h].index_add_(0, idx[h], w[h])
        want_s[h].index_add_(0, idx[h], x[h].double() * w[h][:, None])
    assert torch.equal(counts.double(), want_c)
    torch.testing.assert_close(sums.double(), want_s, rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("H,M,K,D,Q,share", [(1, 20000, 256, 256, 4, False), (2, 5000, 64, 300, 3, False),
                                             (1, 3000, 128, 32, 5, True)])
def test_ema_accumulate_residual_matches_chain(H, M, K, D, Q, share):
    """vq_ema_accumulate_residual_f32: per-stage statistics of the residual stack == rebuilding the chain in float64."""
    from vector_quantization import native

    g = torch.Generator().manual_seed(Q * 100 + K)
    x = torch.randn((H, M, D), generator=g).cuda()
    cb = torch.randn((H, 1 if share else Q, K, D), generator=g).cuda() * 0.5
    r = native.quantize(x, cb, ste=True, stages_share_codebook=share,
                        idx=torch.empty((H, M, Q), dtype=torch.int64, device="cuda"))
    idx = r["idx"]
    counts, sums = native.ema_accumulate_residual(x, cb, idx, ste=True, stages_share_codebook=share)
    torch.cuda.synchronize()
    for h in range(H):
        res = x[h]
        for q in range(Q):
            c = cb[h, 0 if share else q][idx[h, :, q]]
            want_s = torch.zeros((K, D), dtype=torch.float64, device="cuda").index_add_(0, idx[h, :, q], res.double())
            want_c = torch.bincount(idx[h, :, q], minlength=K).double()
            assert torch.equal(counts[h, q].double(), want_c)
            torch.testing.assert_close(sums[h, q].double(), want_s, rtol=1e-5, atol=1e-4)
            quant = res + (c - res)
            res = res - quant


@pytest.mark.parametrize("kind", ["vq", "rvq", "rvq_staged"])
def test_graphed_forward_replays_on_new_data_and_new_weights(oracle, kind):
    """GraphedForward: one hipGraph launch per forward; new inputs and in-place codebook updates need no re-capture."""
    import vector_quantization as vq
    from vector_quantization.codebooks import CodebookParams

    torch.manual_seed(0)
    if kind == "vq":
        mod = vq.VectorQuantize(dim=64, codebook_params=CodebookParams(dim=64, codebook_size=256)).to(DEV).eval()
    elif kind == "rvq":
        mod = vq.ResidualVQ(dim=64, num_quantizers=3, codebook_params=CodebookParams(dim=64, codebook_size=128)).to(DEV).eval()
    else:  # few rows, long sweeps: the launcher runs the stack stage by stage (K-split search + finalize per stage) -- all of it captured
        mod = vq.ResidualVQ(dim=64, num_quantizers=3, codebook_params=CodebookParams(dim=64, codebook_size=4096)).to(DEV).eval()
    fast = vq.GraphedForward(mod, torch.randn(32, 256, 64, device=DEV))
    for step in range(3):
        if step == 2:  # new weights, same buffers
            with torch.no_grad():
                for m in mod.modules():
                    if isinstance(m, vq.Codebook):
                        m.embeddings.copy_(torch.randn_like(m.embeddings))
        x = torch.randn(32, 256, 64, device=DEV)
        q, i, _ = fast(x)
        with torch.no_grad():
            q_ref, i_ref, _ = mod(x)
        assert torch.equal(i, i_ref) and torch.equal(q, q_ref)
    with pytest.raises(ValueError):
        fast(torch.randn(8, 256, 64, device=DEV))
    with pytest.raises(ValueError):
        vq.GraphedForward(mod.train(), torch.randn(32, 256, 64, device=DEV))


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("H,M,K,D", [(1, 4096, 1024, 256), (3, 500, 300, 100), (1, 64, 4096, 512), (2, 1000, 96, 30)])
def test_two_byte_rows_are_widened_in_the_kernel(dtype, H, M, K, D):
    """fp16 / bf16 rows (inference): the kernel's prologue widens them -- the reference's x.float() -- so indices,
    distances and quantized rows are bit-identical to the launch on the pre-widened tensor."""
    from vector_quantization import native

    g = torch.Generator().manual_seed(K + D)
    x = torch.randn((H, M, D), generator=g).to(dtype).cuda()
    cb = torch.randn((H, 1, K, D), generator=g).cuda()
    a = native.quantize(x, cb)
    b = native.quantize(x.float(), cb)
    torch.cuda.synchronize()
    for key in ("idx", "best", "out"):
        assert torch.equal(a[key], b[key]), key
    assert a["out"].dtype == torch.float32
    # strided head views (module layout) and the split-K path take the same route
    xs = torch.randn((M, H, D), generator=g).to(dtype).cuda().permute(1, 0, 2)
    c = native.quantize(xs, cb, flags=native.F_FORCE_SPLIT)
    d = native.quantize(xs.float(), cb)
    torch.cuda.synchronize()
    assert torch.equal(c["idx"], d["idx"]) and torch.equal(c["out"], d["out"])


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_half_precision_module_inference_matches_float_path(dtype):
    import vector_quantization as vq
    from vector_quantization.codebooks import CodebookParams

    torch.manual_seed(0)
    for kw in (dict(dim=64), dict(dim=64, heads=2, codebook_dim=32, separate_codebook_per_head=True), dict(dim=64, heads=2, codebook_dim=32)):
        mod = vq.VectorQuantize(codebook_params=CodebookParams(dim=kw.get("codebook_dim", 64), codebook_size=128), **kw).to(DEV).eval()
        x = torch.randn(4, 100, 64, device=DEV).to(dtype)
        with torch.no_grad():
            q, i, loss = mod(x)
            q2, i2, _ = mod(x.float())
        assert torch.equal(i, i2) and torch.equal(q, q2) and q.dtype == torch.float32


def test_c_abi_rejects_bad_arguments_without_launching():
    """Argument errors come back as negative VQ_E_* codes with a message (include/vq_mi355x.h) -- never a launch."""
    import ctypes

    from vector_quantization import native

    lib = native.load()
    x = torch.randn(1, 64, 32, device=DEV)
    cb = torch.randn(1, 1, 16, 32, device=DEV)
    packed = native.pack_codebooks(cb, 0)
    idx = torch.empty((1, 64, 1), dtype=torch.int64, device=DEV)
    out = torch.empty_like(x)
    ws = torch.empty(int(lib.vq_workspace_bytes(1, 64, 1)), dtype=torch.uint8, device=DEV)

    def args(**over):
        a = native.VqArgs()
        a.H, a.Q, a.M, a.K, a.D, a.metric, a.flags = 1, 1, 64, 16, 32, 0, 0
        a.x, a.x_rs, a.x_hs = x.data_ptr(), 32, 64 * 32
        a.cb, a.cb_hs, a.cb_qs = cb.data_ptr(), 16 * 32, 0
        a.packed, a.pk_hs, a.pk_qs = packed.data_ptr(), packed.shape[-1], 0
        a.out, a.out_rs, a.out_hs = out.data_ptr(), 32, 64 * 32
        a.idx, a.idx_rs, a.idx_hs, a.idx_qs = idx.data_ptr(), 1, 64, 0
        a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel()
        for k, v in over.items():
            setattr(a, k, v)
        return a

    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert lib.vq_quantize_f32(ctypes.byref(args()), stream) == 0
    BADARG, UNSUPPORTED = -1, -2
    for over, want in ((dict(K=0), BADARG), (dict(D=-3), BADARG), (dict(metric=7), BADARG), (dict(idx=None), BADARG),
                       (dict(cb=None), BADARG), (dict(packed=None), BADARG), (dict(workspace_bytes=16), BADARG),
                       (dict(x=None), BADARG), (dict(flags=native.F_X_F16 | native.F_STE), UNSUPPORTED)):
        rc = lib.vq_quantize_f32(ctypes.byref(args(**over)), stream)
        assert rc == want, (over, rc)
        assert lib.vq_last_error(), over
    lse = torch.empty((1, 64), device=DEV)
    assert lib.vq_quantize_lse_f32(ctypes.byref(args(Q=2)), ctypes.c_void_p(lse.data_ptr()), stream) == UNSUPPORTED
    assert lib.vq_nearest_f32(ctypes.byref(args(Q=2)), stream) == BADARG
    assert lib.vq_search_keys_f32(ctypes.byref(args()), 0, None, stream) == BADARG
    assert lib.vq_similarities_f32(ctypes.byref(args()), None, 16, 64 * 16, stream) == BADARG
    assert lib.vq_softmax_stats_f32(ctypes.byref(args()), ctypes.c_float(1.0), None, 0, 0, None, None, stream) == BADARG
    assert lib.vq_packed_floats(0, 32) == 0 and lib.vq_workspace_bytes(0, 10, 1) == 0
    torch.cuda.synchronize()  # nothing faulted
    # the Python binding turns them into exceptions
    with pytest.raises(RuntimeError):
        native.quantize(x, cb, flags=native.F_FORCE_SIMPLE, want_lse=True)
    with pytest.raises(native.NativeUnavailable):
        native.quantize(x.cpu(), cb.cpu())


def test_non_finite_rows_do_not_disturb_their_neighbours():
    """Non-finite inputs are outside the arithmetic contract (ATen's argmax returns the first NaN position, the
    in-lane min tree ignores NaN), but they must stay LOCAL: a row of NaN / inf yields some in-range index, and every other
    row of the same wave, workgroup and launch is unaffected -- for the one-wave and the wave-pair kernels."""
    native = _native()
    for (M, K, D) in ((40000, 300, 64), (40000, 1000, 256), (40000, 200, 384)):
        g = torch.Generator().manual_seed(D)
        x = torch.randn((1, M, D), generator=g).to(DEV)
        cb = torch.randn((1, 1, K, D), generator=g).to(DEV)
        clean = native.quantize(x, cb)
        bad = x.clone()
        bad[0, 5, 3] = float("nan")
        bad[0, 77, :] = float("inf")
        bad[0, 1000, 0] = float("-inf")
        r = native.quantize(bad, cb)
        torch.cuda.synchronize()
        idx = r["idx"][0, :, 0]
        assert int(idx.min()) >= 0 and int(idx.max()) < K
        keep = torch.ones(M, dtype=torch.bool, device=DEV)
        keep[[5, 77, 1000]] = False
        assert torch.equal(idx[keep], clean["idx"][0, :, 0][keep])
        assert torch.equal(r["out"][0][keep], clean["out"][0][keep])


@pytest.mark.parametrize("M,K,D", [(1_200_000, 64, 1024), (9_000_000, 40, 256)])
def test_inputs_beyond_2_31_elements(M, K, D):
    """Row offsets past 2^31 elements (9.2 GB of rows; 4.9 GB of wide rows through several row chunks): every pointer offset
    in the launchers and kernels is 64-bit.  Checked against the one-thread-per-row kernel on every row."""
    native = _native()
    g = torch.Generator(device=DEV).manual_seed(1)
    x = torch.randn((1, M, D), device=DEV, generator=g)
    cb = torch.randn((1, 1, K, D), device=DEV, generator=g)
    a = native.quantize(x, cb)
    s = native.quantize(x, cb, flags=native.F_FORCE_SIMPLE)
    torch.cuda.synchronize()
    assert torch.equal(a["idx"], s["idx"])
    assert torch.equal(a["best"].view(torch.int32), s["best"].view(torch.int32))
    assert torch.equal(a["out"], s["out"])
    del x, a, s
    torch.cuda.empty_cache()


@pytest.mark.parametrize("kind", ["vq", "rvq", "grvq"])
def test_eager_forward_right_after_graph_capture(oracle, kind):
    """the capturing forward refills the packed-image caches with tensors from the graph's private pool whose pack
    kernels were only recorded.  An eager module(x) BEFORE the first replay (e.g. the fallback for another batch shape) must
    not search that uninitialised memory."""
    import vector_quantization as vq
    from vector_quantization.codebooks import CodebookParams

    torch.manual_seed(1)
    if kind == "vq":
        mod = vq.VectorQuantize(dim=64, codebook_params=CodebookParams(dim=64, codebook_size=256))
    elif kind == "rvq":
        mod = vq.ResidualVQ(dim=64, num_quantizers=3, codebook_params=CodebookParams(dim=64, codebook_size=128))
    else:
        mod = vq.GroupedResidualVQ(dim=64, groups=2, num_quantizers=2, codebook_params=CodebookParams(dim=32, codebook_size=64))
    mod = mod.to(DEV).eval()
    fast = vq.GraphedForward(mod, torch.randn(8, 64, 64, device=DEV))
    x = torch.randn(5, 40, 64, device=DEV)  # another shape: eager fallback, no replay has happened yet
    with torch.no_grad():
        q, idx, _ = mod(x)
    torch.cuda.synchronize()
    # the codes the search saw must be the module's codes: every output row is the sum of codebook rows at its indices
    if kind == "vq":
        want = mod._codebook.embeddings[0][idx]
    elif kind == "rvq":
        want = sum(layer._codebook.embeddings[0][idx[..., i]] for i, layer in enumerate(mod.layers))
    else:
        want = torch.cat([sum(layer._codebook.embeddings[0][idx[g][..., i]] for i, layer in enumerate(rvq.layers))
                          for g, rvq in enumerate(mod.rvqs)], dim=-1)
    torch.testing.assert_close(q, want, rtol=0, atol=1e-6)
    flat = x.reshape(-1, 64).cpu().numpy()
    if kind == "vq":
        ri, _ = oracle.nearest(flat, mod._codebook.embeddings[0].cpu().numpy(), oracle.EUCLID)
        assert np.array_equal(idx.reshape(-1).cpu().numpy(), ri)
    q2, i2, _ = fast(torch.randn(8, 64, 64, device=DEV))  # and the graph still replays
    assert torch.isfinite(q2).all()


@pytest.mark.parametrize("kind", ["vq", "rvq", "grvq"])
def test_eval_forward_under_inference_mode(kind):
    """torch.inference_mode() is the standard serving idiom; fresh modules, first call inside it."""
    import vector_quantization as vq
    from vector_quantization.codebooks import CodebookParams

    torch.manual_seed(2)
    if kind == "vq":
        mod = vq.VectorQuantize(dim=64, codebook_params=CodebookParams(dim=64, codebook_size=256))
    elif kind == "rvq":
        mod = vq.ResidualVQ(dim=64, num_quantizers=3, codebook_params=CodebookParams(dim=64, codebook_size=128))
    else:
        mod = vq.GroupedResidualVQ(dim=64, groups=2, num_quantizers=2, codebook_params=CodebookParams(dim=32, codebook_size=64))
    mod = mod.to(DEV).eval()
    x = torch.randn(4, 50, 64, device=DEV)
    with torch.inference_mode():
        q1, i1, l1 = mod(x)
        q2, i2, l2 = mod(x)
    with torch.no_grad():
        q3, i3, l3 = mod(x)
    assert torch.equal(i1, i2) and torch.equal(i1, i3) and torch.equal(q1, q3)
    assert float(l1.sum()) == 0.0 and float(l3.sum()) == 0.0
