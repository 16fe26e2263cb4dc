"""GPU: the drop-in modules on cuda:0 (native HIP backend, no checker installed) against the golden vectors
captured from the reference, plus module-level behaviours (autograd, EMA bookkeeping, state_dict names)."""
from __future__ import annotations

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from build_case import build  # noqa: E402
from cases import CASES, LOSS_CASES  # noqa: E402
from check_case import compare, compare_loss  # noqa: E402
from helpers import load_golden  # noqa: E402


@pytest.fixture(autouse=True)
def _native_backend():
    from vector_quantization import native, search

    native.load()
    search.set_backend(None)
    assert search.get_backend().name == "hip-gfx950"
    yield


GPU_CASES = [c for c in CASES if not c.get("cpu_only", False)]  # (host-RNG-dependent bookkeeping is pinned on the CPU)


@pytest.mark.parametrize("case", GPU_CASES, ids=[c["name"] for c in GPU_CASES])
def test_module_matches_reference_golden_on_gpu(case):
    arrays, meta = load_golden(case["name"])
    mod, x, kwargs, cb = build(case, arrays, device="cuda:0")
    with torch.no_grad():
        outputs = mod(x, **kwargs)
    torch.cuda.synchronize()
    compare(case, arrays, meta, outputs, x, cb, mod)


SEEDED_CASES = [c for c in CASES if c.get("cpu_only", False)]


@pytest.mark.parametrize("case", SEEDED_CASES, ids=[c["name"] for c in SEEDED_CASES])
def test_kmeans_seeding_and_dead_code_reseeding_match_reference_golden_on_gpu(case, monkeypatch):
    """SURVEY 8f rank 2 on the device: k-means seeding (Lloyd iterations = native search + native
    vq_ema_accumulate_f32) and dead-code re-seeding against the reference's state.  The goldens were produced with torch's
    CPU generator (torch.manual_seed(forward_seed) before the forward), whose stream a device generator cannot reproduce:
    the row INDICES are drawn with the CPU generator exactly as the CPU-pinned test draws them and are then used to index
    the rows on the GPU, so everything after the draw -- assignment, per-cluster sums, the codebook writes -- is the
    product's device path."""
    from vector_quantization import codebook as codebook_mod

    def pick_rows_with_cpu_draws(rows, count):
        n = rows.shape[0]
        sel = torch.randperm(n)[:count] if n >= count else torch.randint(0, n, (count,))
        return rows[sel.to(rows.device)]

    monkeypatch.setattr(codebook_mod, "_pick_rows", pick_rows_with_cpu_draws)
    arrays, meta = load_golden(case["name"])
    mod, x, kwargs, cb = build(case, arrays, device="cuda:0")
    torch.manual_seed(case["forward_seed"])
    with torch.no_grad():
        outputs = mod(x, **kwargs)
    torch.cuda.synchronize()
    assert mod._codebook.embeddings.is_cuda
    compare(case, arrays, meta, outputs, x, cb, mod)


@pytest.mark.parametrize("case", LOSS_CASES, ids=[c["name"] for c in LOSS_CASES])
def test_similarity_consuming_losses_match_reference_golden_on_gpu(case):
    """SURVEY 8f rank 3 on the native kernels (vq_softmax_stats_f32 forward, vq_similarities_f32 chunks backward)."""
    arrays, meta = load_golden(case["name"])
    mod, x, kwargs, _ = build(case, arrays, device="cuda:0")
    compare_loss(case, arrays, meta, mod, x, kwargs)


@pytest.mark.parametrize("name", ["ce_commit", "ce_commit_cfg2", "ce_indices_ignore", "div_t1", "div_mh_shared"])
def test_losses_are_chunk_size_independent_on_gpu(name, monkeypatch):
    from cases import LOSS_CASES_BY_NAME
    from vector_quantization import losses

    monkeypatch.setattr(losses, "CHUNK_BYTES", 1)  # 128-row chunks
    case = LOSS_CASES_BY_NAME[name]
    arrays, meta = load_golden(name)
    mod, x, kwargs, _ = build(case, arrays, device="cuda:0")
    compare_loss(case, arrays, meta, mod, x, kwargs)


def test_cross_entropy_commitment_full_size_memory_bound():
    """cfg2 rows (M = 262144, K = 1024): forward + backward of the cross-entropy commitment loss without ever holding
    the 1 GiB similarity matrix: peak extra memory stays near the chunk bound, gradient equals a chunked torch reference."""
    import vector_quantization as vq
    from vector_quantization import losses
    from vector_quantization.codebooks import CodebookParams

    dev = "cuda:0"
    torch.manual_seed(0)
    mod = vq.VectorQuantize(dim=256, codebook_params=CodebookParams(dim=256, codebook_size=1024),
                            commitment_use_cross_entropy_loss=True).to(dev).train()
    x = torch.randn(256, 1024, 256, device=dev, requires_grad=True)
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    q, idx, loss = mod(x, freeze_codebook=True)
    loss.sum().backward()
    torch.cuda.synchronize()
    peak = torch.cuda.max_memory_allocated() - base
    # outputs (q 256 MiB, idx) + x.grad (256 MiB) + a few chunk-sized temporaries; the full matrix alone would be 1 GiB
    # for sims plus as much again for the softmax and its gradient
    assert peak < (256 + 256 + 64) * 2**20 + 6 * losses.CHUNK_BYTES, peak / 2**20
    codes = mod._codebook.embeddings.detach()
    for b0 in (0, 131, 255):
        xs = x.detach()[b0].clone().requires_grad_(True)
        ce = torch.nn.functional.cross_entropy(-torch.cdist(xs[None], codes)[0], idx[b0], reduction="sum") / (256 * 1024)
        ce.backward()
        torch.testing.assert_close(x.grad[b0], xs.grad, rtol=1e-3, atol=1e-9)


def test_gpu_equals_oracle_backend_on_all_cases(oracle):
    """Same modules, same inputs: native backend on the GPU vs checker backend on the CPU -> identical idx / q."""
    from helpers import OracleBackend
    from vector_quantization import search

    for case in GPU_CASES:
        if case["name"] in ("cfg5_S",):
            continue
        arrays, _ = load_golden(case["name"])
        mod, x, kwargs, _ = build(case, arrays, device="cuda:0")
        with torch.no_grad():
            got = mod(x, **kwargs)
        search.set_backend(OracleBackend)
        try:
            mod_c, x_c, kwargs_c, _ = build(case, arrays, device="cpu")
            with torch.no_grad():
                ref = mod_c(x_c, **kwargs_c)
        finally:
            search.set_backend(None)
        np.testing.assert_array_equal(got[1].cpu().numpy(), ref[1].numpy(), err_msg=case["name"])
        exact = (not mod.__dict__.get("has_projections", False) and not case.get("codebook_dim") and case["name"] != "proj_mh"
                 and not (case["training"] and case.get("transform_input") == "l2norm")  # F.normalize differs GPU vs CPU
                 # a shared codebook rewritten by EMA between the stages: its sums are float atomics on the GPU
                 and not (case.get("shared_codebook") and case["training"] and not case.get("freeze_codebook", True)))
        if exact:
            np.testing.assert_array_equal(got[0].cpu().numpy(), ref[0].numpy(), err_msg=case["name"])
        else:
            np.testing.assert_allclose(got[0].cpu().numpy(), ref[0].numpy(), atol=1e-5, err_msg=case["name"])
        np.testing.assert_allclose(got[2].cpu().numpy(), ref[2].numpy(), atol=1e-6, err_msg=case["name"])


def test_autograd_straight_through_and_commitment():
    import vector_quantization as vq
    from vector_quantization.codebooks import CodebookParams

    torch.manual_seed(0)
    dev = "cuda:0"
    mod = vq.VectorQuantize(dim=32, codebook_params=CodebookParams(dim=32, codebook_size=64), commitment_weight=0.25).to(dev)
    mod.train()
    x = torch.randn(3, 17, 32, device=dev, requires_grad=True)
    q, idx, loss = mod(x, freeze_codebook=True)
    w = torch.randn_like(q)
    ((q * w).sum() + 3.0 * loss.sum()).backward()
    codes = mod._codebook.embeddings[0][idx]
    expect = w + 3.0 * 0.25 * 2.0 * (x.detach() - codes) / x.numel()
    torch.testing.assert_close(x.grad, expect, rtol=1e-5, atol=1e-6)


def test_ema_update_matches_one_hot_formulation():
    """Training-mode bookkeeping after the hot path: same statistics as the reference's one-hot products."""
    import vector_quantization as vq
    from vector_quantization.codebooks import CodebookParams

    torch.manual_seed(1)
    dev = "cuda:0"
    K, D = 32, 16
    mod = vq.VectorQuantize(dim=D, codebook_params=CodebookParams(dim=D, codebook_size=K, threshold_ema_dead_code=0,
                                                                decay=0.8)).to(dev)
    with torch.no_grad():
        mod._codebook.embeddings.copy_(torch.randn(1, K, D))
        mod._codebook.embed_avg.copy_(mod._codebook.embeddings)
    emb0 = mod._codebook.embeddings.clone()
    avg0 = mod._codebook.embed_avg.clone()
    mod.train()
    x = torch.randn(4, 50, D, device=dev)
    with torch.no_grad():
        _, idx, _ = mod(x)
    onehot = torch.nn.functional.one_hot(idx.reshape(-1), K).float()
    cs = torch.zeros(1, K, device=dev).lerp(onehot.sum(0)[None], 0.2)
    avg = avg0.lerp((onehot.t() @ x.reshape(-1, D))[None], 0.2)
    tot = cs.sum(-1, keepdim=True)
    sm = (cs + 1e-5) / (tot + K * 1e-5) * tot
    torch.testing.assert_close(mod._codebook.cluster_size, cs, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(mod._codebook.embed_avg, avg, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(mod._codebook.embeddings, avg / sm[..., None], rtol=1e-4, atol=1e-5)
    assert not torch.equal(emb0, mod._codebook.embeddings)


def test_state_dict_names_are_the_reference_checkpoint_format():
    import vector_quantization as vq
    from vector_quantization.codebooks import CodebookParams

    mod = vq.VectorQuantize(dim=16, codebook_params=CodebookParams(dim=16, codebook_size=8))
    assert set(mod.state_dict().keys()) == {"_codebook.cluster_size", "_codebook.embed_avg", "_codebook.embeddings"}
    rvq = vq.ResidualVQ(dim=16, num_quantizers=2, codebook_params=CodebookParams(dim=16, codebook_size=8))
    assert "layers.1._codebook.embeddings" in rvq.state_dict()


def test_kmeans_init_runs_on_native_search():
    import vector_quantization as vq
    from vector_quantization.codebooks import CodebookParams, KmeansParameters

    torch.manual_seed(2)
    dev = "cuda:0"
    mod = vq.VectorQuantize(dim=16, codebook_params=CodebookParams(dim=16, codebook_size=8, initialization_by_kmeans=True,
                                                                 kmeans_params=KmeansParameters(iter=5))).to(dev)
    mod.train()
    centers = torch.randn(8, 16, device=dev) * 5
    x = (centers[torch.randint(0, 8, (4, 100), device=dev)] + 0.01 * torch.randn(4, 100, 16, device=dev))
    q, idx, loss = mod(x)
    assert q.shape == x.shape and idx.shape == x.shape[:-1]
    assert mod._codebook.is_initialized
    assert float(mod._codebook.cluster_size.sum()) > 0


def test_learnable_codebook_gradients_match_torch_formulation():
    """learnable_codebook (no EMA): the codebook receives the commitment-loss gradient through the gathered rows
    (reference: one-hot einsum + mse_loss(quantize, x), vector_quantize_pytorch.py:263-269,362)."""
    import vector_quantization as vq
    from vector_quantization.codebooks import CodebookParams

    torch.manual_seed(3)
    dev = "cuda:0"
    K, D = 16, 8
    mod = vq.VectorQuantize(dim=D, codebook_params=CodebookParams(dim=D, codebook_size=K, learnable_codebook=True,
                                                                ema_update=False), commitment_weight=0.5).to(dev)
    mod.train()
    x = torch.randn(4, 9, D, device=dev, requires_grad=True)
    q, idx, loss = mod(x)
    loss.sum().backward()
    codes = mod._codebook.embeddings.detach()[0]
    sel = codes[idx]
    n = x.numel()
    # d/dx of w * mean((c - x)^2) and d/dc scattered onto the selected rows
    torch.testing.assert_close(x.grad, 0.5 * 2.0 * (x.detach() - sel) / n, rtol=1e-5, atol=1e-7)
    expect = torch.zeros_like(codes)
    expect.index_add_(0, idx.reshape(-1), (0.5 * 2.0 * (sel - x.detach()) / n).reshape(-1, D))
    torch.testing.assert_close(mod._codebook.embeddings.grad[0], expect, rtol=1e-5, atol=1e-7)


def test_similarities_on_demand_and_codebook_forward_api():
    from vector_quantization.codebooks import Codebook

    torch.manual_seed(4)
    dev = "cuda:0"
    cb = Codebook(dim=16, codebook_size=32, num_codebooks=2).to(dev).eval()
    x = torch.randn(2, 3, 11, 16, device=dev)  # [h, b, n, d]
    q, ind, sims = cb(x)  # like the reference: the third return value is the full similarity tensor
    assert q.shape == x.shape and ind.shape == (2, 3, 11) and sims.shape == (2, 3, 11, 32)
    assert torch.equal(ind, sims.argmax(-1))
    q2, ind2, none = cb(x, return_similarities=False)
    assert none is None and torch.equal(ind2, ind) and torch.equal(q2, q)
    x3 = torch.randn(3, 11, 16, device=dev)  # [b, n, d] with a single codebook
    cb1 = Codebook(dim=16, codebook_size=32).to(dev).eval()
    q3, ind3, _ = cb1(x3)
    assert q3.shape == x3.shape and ind3.shape == (3, 11)


def test_multihead_mask_train_matches_manual():
    import vector_quantization as vq
    from vector_quantization.codebooks import CodebookParams

    torch.manual_seed(5)
    dev = "cuda:0"
    mod = vq.VectorQuantize(dim=32, heads=2, codebook_dim=16, separate_codebook_per_head=True,
                            codebook_params=CodebookParams(dim=16, codebook_size=24)).to(dev).train()
    x = torch.randn(3, 10, 32, device=dev)
    mask = torch.zeros(3, 10, dtype=torch.bool, device=dev)
    mask[0, :10] = True
    mask[1, :4] = True
    mask[2, :1] = True
    with torch.no_grad():
        q, idx, loss = mod(x, mask=mask, freeze_codebook=True)
    codes = mod._codebook.embeddings
    xh = x.view(3, 10, 2, 16)
    sel = torch.stack([codes[h][idx[..., h]] for h in range(2)], dim=2)  # [b, n, h, d]
    manual = ((sel - xh) ** 2)[mask].mean()
    torch.testing.assert_close(loss[0], manual, rtol=1e-5, atol=1e-7)
    assert torch.equal(q[~mask], x[~mask])
    torch.testing.assert_close(q[mask], (xh + (sel - xh)).reshape(3, 10, 32)[mask], rtol=0, atol=1e-6)


class _TorchBackwardBackend:
    """The native backend minus the fused backward kernels: autograd then takes the PyTorch formulas in search.py /
    losses.py -- the reference implementation the native backward passes are compared with."""
    name = "hip-gfx950 (torch backward)"

    def __init__(self):
        from vector_quantization import search

        native_backend = search._NativeBackend
        self.quantize = native_backend.quantize
        self.similarities = native_backend.similarities
        self.softmax_stats = native_backend.softmax_stats
        self.ema_accumulate = native_backend.ema_accumulate
        self.ema_accumulate_residual = native_backend.ema_accumulate_residual
        self.ema_update = native_backend.ema_update


@pytest.mark.parametrize("kind", ["vq", "vq_eval_learnable_off", "vq_heads_sep", "vq_heads_shared", "rvq", "rvq_shared", "grvq"])
def test_native_quantize_backward_equals_torch_formulas(kind):
    """vq_quantize_backward_f32 (one pass) vs the gather + elementwise autograd formulas, on the same modules."""
    import vector_quantization as vq
    from vector_quantization import search
    from vector_quantization.codebooks import CodebookParams

    dev = "cuda:0"
    torch.manual_seed(0)
    if kind.startswith("vq"):
        kw = dict(vq_heads_sep=dict(dim=64, heads=2, codebook_dim=32, separate_codebook_per_head=True),
                  vq_heads_shared=dict(dim=64, heads=2, codebook_dim=32)).get(kind, dict(dim=64))
        mod = vq.VectorQuantize(codebook_params=CodebookParams(dim=kw.get("codebook_dim", 64), codebook_size=128),
                                commitment_weight=0.7, **kw).to(dev)
        x = torch.randn(3, 50, 64, device=dev)
    elif kind == "grvq":
        mod = vq.GroupedResidualVQ(dim=64, groups=2, num_quantizers=3,
                                   codebook_params=CodebookParams(dim=32, codebook_size=64)).to(dev)
        x = torch.randn(3, 50, 64, device=dev)
    else:
        mod = vq.ResidualVQ(dim=48, num_quantizers=4, shared_codebook=(kind == "rvq_shared"),
                            codebook_params=CodebookParams(dim=48, codebook_size=96)).to(dev)
        x = torch.randn(3, 50, 48, device=dev)
    mod.train()
    if kind == "vq_eval_learnable_off":
        mod.eval()
    w = torch.randn_like(x)

    def grads(backend):
        search.set_backend(backend)
        try:
            xs = x.clone().requires_grad_(True)
            out = mod(xs, freeze_codebook=True) if mod.training else mod(xs)
            objective = (out[0] * w).sum() + out[2].sum() * 3.0
            if not objective.requires_grad:
                return torch.zeros_like(xs)
            objective.backward()
            return xs.grad if xs.grad is not None else torch.zeros_like(xs)
        finally:
            search.set_backend(None)

    got = grads(None)
    want = grads(_TorchBackwardBackend())
    torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-6 * float(want.abs().max() + 1e-30))
