"""GPU: rows WIDER than 512 dims (the reference has no width limit: cdist / einsum over any D, codebooks.py:122-129,386).

Such rows are swept in 512-dim slices whose distance chains wait in the workspace between two launches
(vq_search_pair512<metric, false, 0, WIDE> -- vq_search_mfma<.., WIDE> for short sweeps and for the last, narrower slice --
csrc/vq_kernels.hip run_wide); the chain of a (row, code)
pair is still the oracle's k-ordered fmaf chain, so indices AND winning distances must be bit-identical to the CPU oracle.
Covered: D just above 512 / not a multiple of 4 / several slices, K below one sub-tile and across two code chunks
(4096 codes each), split-K launches (few row blocks) and unsplit ones (many), several row chunks (heads shrink the chunk),
straight-through + squared error, both metrics, duplicated codebooks (ties), the sharded keys path and the modules.
The one-thread-per-row kernel (VQ_F_FORCE_SIMPLE: same chain, no MFMA) is the witness where the oracle would take minutes."""
from __future__ import annotations

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from gen import make_codebook, make_x  # noqa: E402

DEV = "cuda:0"


def _native():
    from vector_quantization import native

    native.load()
    return native


SHAPES = [
    # (H, M, K, D)
    (1, 300, 1000, 520),
    (1, 257, 333, 1024),
    (2, 100, 64, 640),
    (1, 129, 40, 2048),
    (1, 64, 1, 513),
    (1, 33, 4100, 516),      # two code chunks, the second holds 4 codes
    (1, 5000, 96, 768),
    (1, 40000, 32, 600),     # more row blocks than CUs: no K split
    (3, 70, 7, 1537),        # four slices, the last one a single dim
    (1, 200, 500, 560),      # last slice padded to 64 dims (four sub-tiles per staged tile)
]


@pytest.mark.parametrize("H,M,K,D", SHAPES)
@pytest.mark.parametrize("cls", ["S", "G", "Gdup", "R"])
@pytest.mark.parametrize("metric", [0, 1])
def test_wide_rows_bit_exact(oracle, H, M, K, D, cls, metric):
    native = _native()
    x = make_x((H, M, D), cls)
    cb = make_codebook(H, K, D, cls)
    if cls != "S" and M * K * D > 3e8:
        # oracle time: the large shapes meet the CPU oracle once (class S); the other data classes are checked against the
        # scalar kernel (one thread per row, the same k-ordered chain: equal bits) on every row
        got = native.quantize(x.to(DEV), cb[:, None].contiguous().to(DEV), metric=metric, want_sq_err=True)
        wit = native.quantize(x.to(DEV), cb[:, None].contiguous().to(DEV), metric=metric, want_sq_err=True, flags=native.F_FORCE_SIMPLE)
        assert torch.equal(got["idx"], wit["idx"]) and torch.equal(got["out"], wit["out"])
        assert torch.equal(got["best"].view(torch.int32), wit["best"].view(torch.int32)), "distances differ"
        torch.testing.assert_close(got["sq_err"], wit["sq_err"], rtol=1e-6, atol=0)
        if cls == "Gdup" and K >= 2 and K % 2 == 0:
            assert int(got["idx"].max()) < max(K // 2, 1)
        return
    ref = oracle.vq_forward(x.numpy(), cb.numpy(), metric, training=False)
    got = native.quantize(x.to(DEV), cb[:, None].contiguous().to(DEV), metric=metric, want_sq_err=True)
    idx = got["idx"][..., 0].cpu().numpy()
    np.testing.assert_array_equal(idx, ref["idx"])
    assert np.array_equal(got["best"][..., 0].cpu().numpy().view(np.uint32), ref["best"].view(np.uint32)), "distances differ"
    np.testing.assert_array_equal(got["out"].cpu().numpy(), ref["out"])
    np.testing.assert_allclose(got["sq_err"].cpu().numpy()[0], ref["sq_err"], rtol=1e-6)
    if cls == "Gdup" and K >= 2 and K % 2 == 0:
        assert idx.max() < max(K // 2, 1)


@pytest.mark.parametrize("H,M,K,D", [(1, 300, 1000, 520), (2, 100, 64, 640)])
@pytest.mark.parametrize("metric", [0, 1])
def test_wide_rows_training_outputs(oracle, H, M, K, D, metric):
    native = _native()
    x = make_x((H, M, D), "S")
    cb = make_codebook(H, K, D, "S")
    ref = oracle.vq_forward(x.numpy(), cb.numpy(), metric, training=True)
    got = native.quantize(x.to(DEV), cb[:, None].contiguous().to(DEV), metric=metric, want_sq_err=True, ste=True)
    np.testing.assert_array_equal(got["idx"][..., 0].cpu().numpy(), ref["idx"])
    np.testing.assert_array_equal(got["out"].cpu().numpy(), ref["out"])
    np.testing.assert_allclose(got["sq_err"].cpu().numpy()[0], ref["sq_err"], rtol=1e-6)


@pytest.mark.parametrize("H,M,K,D,metric", [
    (8, 4200, 4100, 520, 0),    # heads shrink the row chunk to 4096 rows: two row chunks x two code chunks
    (1, 70000, 300, 1030, 0),   # one chunk, 547 row blocks, three slices
    (2, 9000, 5000, 777, 1),
    (1, 1460, 5000, 2056, 0),   # five slices, K split 22 ways, two code chunks: the |x|^2 chain is read and written by every split
    (1, 20000, 1000, 1600, 0),
    (2, 40000, 1000, 768, 0),   # enough row blocks and one code chunk: the last slice finishes the inference call itself
    (1, 66000, 64, 520, 1),     # (no keys, no finalize kernel), also with a ragged last row block
])
def test_wide_rows_chunked_equals_scalar_kernel(oracle, H, M, K, D, metric):
    native = _native()
    g = torch.Generator().manual_seed(M + K + D)
    x = torch.randn((H, M, D), generator=g).to(DEV)
    cb = torch.randn((H, 1, K, D), generator=g).to(DEV)
    r = native.quantize(x, cb, metric=metric)
    s = native.quantize(x, cb, metric=metric, flags=native.F_FORCE_SIMPLE)
    assert torch.equal(r["idx"], s["idx"])
    assert torch.equal(r["best"].view(torch.int32), s["best"].view(torch.int32))
    assert torch.equal(r["out"], s["out"])
    rows = torch.cat([torch.randperm(M, generator=torch.Generator().manual_seed(1))[:24], torch.arange(M - 8, M)])
    for h in range(0, H, 3):
        ri, rb = oracle.nearest(x[h, rows].cpu().numpy(), cb[h, 0].cpu().numpy(), metric)
        np.testing.assert_array_equal(r["idx"][h, rows, 0].cpu().numpy(), ri)
        assert np.array_equal(r["best"][h, rows, 0].cpu().numpy().view(np.uint32), rb.view(np.uint32))
    # searching the quantized rows again returns them (exact gathers of codebook rows)
    again = native.quantize(r["out"], cb, metric=metric) if metric == 0 else None
    if again is not None:
        assert torch.equal(again["out"], r["out"])


def test_wide_rows_strided_views_and_unaligned_dims():
    """Row strides larger than D (a view into a wider buffer) and D % 4 != 0 (scalar loads in the prologue)."""
    native = _native()
    g = torch.Generator().manual_seed(3)
    M, K = 500, 200
    for D, pad in ((770, 6), (1025, 3), (644, 0)):
        buf = torch.randn((1, M, D + pad), generator=g).to(DEV)
        x = buf[..., :D]
        cb = torch.randn((1, 1, K, D), generator=g).to(DEV)
        a = native.quantize(x, cb)
        b = native.quantize(x.contiguous(), cb, flags=native.F_FORCE_SIMPLE)
        assert torch.equal(a["idx"], b["idx"]) and torch.equal(a["best"].view(torch.int32), b["best"].view(torch.int32))
        assert torch.equal(a["out"], b["out"])


def test_wide_rows_sharded_keys(oracle):
    """The keys path of a K-sharded codebook (vq_search_keys_f32 with an index offset) at D > 512."""
    native = _native()
    H, M, K, D = 1, 700, 600, 900
    x = make_x((H, M, D), "S").to(DEV)
    cb = make_codebook(H, K, D, "S").to(DEV)
    keys = torch.empty((H, M), dtype=torch.int64, device=DEV)
    native.keys_init(keys)
    for lo, hi in ((0, 250), (250, 600)):
        native.search_keys(x, cb[:, lo:hi].contiguous(), keys, idx_offset=lo)
    r = native.finalize_keys(x, cb, keys)
    ref = oracle.vq_forward(x.cpu().numpy(), cb.cpu().numpy(), 0, training=False)
    np.testing.assert_array_equal(r["idx"].cpu().numpy(), ref["idx"])
    assert np.array_equal(r["best"].cpu().numpy().view(np.uint32), ref["best"].view(np.uint32))
    np.testing.assert_array_equal(r["out"].cpu().numpy(), ref["out"])


def test_wide_rows_workspace_is_checked():
    """The C ABI refuses a workspace without room for the chains (and says which function sizes it)."""
    import ctypes

    native = _native()
    lib = native.load()
    H, M, K, D = 1, 256, 64, 600
    assert lib.vq_workspace_bytes_wide(H, M, K, D) > lib.vq_workspace_bytes(H, M, 1)
    assert lib.vq_workspace_bytes_wide(H, M, K, 512) == lib.vq_workspace_bytes(H, M, 1)
    assert lib.vq_packed_floats(K, D) == lib.vq_packed_floats(K, 512) + lib.vq_packed_floats(K, D - 512)  # slices of 512 + 88 dims
    x = torch.randn((H, M, D), device=DEV)
    cb = torch.randn((H, 1, K, D), device=DEV)
    packed = native.pack_codebooks(cb, 0)
    idx = torch.empty((H, M, 1), dtype=torch.int64, device=DEV)
    ws = torch.empty(int(lib.vq_workspace_bytes(H, M, 1)), dtype=torch.uint8, device=DEV)
    a = native.VqArgs()
    a.H, a.Q, a.M, a.K, a.D, a.metric, a.flags = H, 1, M, K, D, 0, 0
    a.x, a.x_rs, a.x_hs = x.data_ptr(), D, M * D
    a.cb, a.cb_hs, a.cb_qs = cb.data_ptr(), K * D, K * D
    a.packed, a.pk_hs, a.pk_qs = packed.data_ptr(), packed.shape[-1], packed.shape[-1]
    a.idx, a.idx_rs, a.idx_hs, a.idx_qs = idx.data_ptr(), 1, M, 1
    a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel()
    rc = lib.vq_quantize_f32(ctypes.byref(a), None)
    assert rc == -1 and b"vq_workspace_bytes_wide" in lib.vq_last_error()
    torch.cuda.synchronize()


def test_wide_rows_modules(oracle):
    """VectorQuantize / ResidualVQ at dim > 512 against the same modules on the CPU checker backend."""
    import vector_quantization as vq
    from helpers import OracleBackend
    from vector_quantization import search
    from vector_quantization.codebooks import CodebookParams

    torch.manual_seed(0)
    cases = (
        (768, lambda: vq.VectorQuantize(dim=768, codebook_params=CodebookParams(dim=768, codebook_size=200))),
        (640, lambda: vq.ResidualVQ(dim=640, num_quantizers=3, codebook_params=CodebookParams(dim=640, codebook_size=64))),
    )
    for d, make in cases:
        xin = torch.randn(2, 50, d)
        torch.manual_seed(1)
        ref_mod = make().eval()
        search.set_backend(OracleBackend)
        try:
            rq, ri, _ = ref_mod(xin)
        finally:
            search.set_backend(None)
        torch.manual_seed(1)
        mod = make().eval().to(DEV)
        q, i, _ = mod(xin.to(DEV))
        assert torch.equal(i.cpu(), ri)
        torch.testing.assert_close(q.cpu(), rq, rtol=0, atol=1e-6)


def test_wide_rows_graph_replay_and_compile():
    """The multi-launch sequence of a wide-row forward is capturable (no host synchronisation, workspace from the capture's
    pool) and traceable (torch.compile goes through torch.ops.vq_mi355x.quantize_into like the narrow path)."""
    import vector_quantization as vq
    from vector_quantization.codebooks import CodebookParams

    torch.manual_seed(0)
    mod = vq.VectorQuantize(dim=768, codebook_params=CodebookParams(dim=768, codebook_size=300)).to(DEV).eval()
    fast = vq.GraphedForward(mod, torch.randn(4, 200, 768, device=DEV))
    for step in range(3):
        if step == 2:
            with torch.no_grad():
                mod._codebook.embeddings.copy_(torch.randn_like(mod._codebook.embeddings))
        x = torch.randn(4, 200, 768, device=DEV)
        q, i, _ = fast(x)
        with torch.no_grad():
            q_ref, i_ref, _ = mod(x)
        assert torch.equal(i, i_ref) and torch.equal(q, q_ref)
    cmod = torch.compile(mod, fullgraph=True)
    x = torch.randn(4, 200, 768, device=DEV)
    with torch.no_grad():
        q, i, _ = cmod(x)
        q_ref, i_ref, _ = mod(x)
    assert torch.equal(i, i_ref) and torch.equal(q, q_ref)


@pytest.mark.parametrize("H,M,K,D", [(1, 300, 1000, 520), (2, 100, 64, 640), (1, 77, 4100, 516), (1, 5000, 96, 768),
                                     (3, 70, 7, 1537), (1, 40000, 33, 600)])
@pytest.mark.parametrize("metric", [0, 1])
def test_wide_rows_similarities_bit_exact(oracle, H, M, K, D, metric):
    """The similarity matrix (third return value of Codebook.forward, codebooks.py:386,435) at D > 512: the sliced sweep whose
    last slice writes -sqrt / dot instead of reducing.  Equal to the scalar kernel on every entry, to the oracle's winning
    values where the search looks, and the row argmax is the search's index."""
    native = _native()
    g = torch.Generator().manual_seed(M + K + D)
    x = torch.randn((H, M, D), generator=g).to(DEV)
    cb = torch.randn((H, K, D), generator=g).to(DEV)
    sims = native.similarities(x, cb, metric=metric)
    ref = native.similarities(x, cb, metric=metric, flags=native.F_FORCE_SIMPLE)
    assert torch.equal(sims.view(torch.int32), ref.view(torch.int32))
    r = native.quantize(x, cb[:, None].contiguous(), metric=metric)
    best = sims.max(dim=-1).values
    want = -r["best"][..., 0] if metric == 0 else r["best"][..., 0]
    assert torch.equal(best.view(torch.int32), want.view(torch.int32))
    # strided destination (a column window of a wider matrix)
    wide = torch.full((H, M, K + 8), 7.0, device=DEV)
    native.similarities(x, cb, metric=metric, out=wide[..., 4:4 + K])
    assert torch.equal(wide[..., 4:4 + K], sims) and bool((wide[..., :4] == 7).all()) and bool((wide[..., 4 + K:] == 7).all())
