"""API-conformance matrix modelled on the reference's own test strategy: every module is
built as in the reference's tests and called -- in TRAINING mode, as the reference's tests do -- on a series, an
image and a video, channel-last and channel-first; shapes/dtypes of the three returns are asserted.  Runs on the
CPU with the checker backend and (marked gpu) on the device with the native backend."""
from __future__ import annotations

import pytest
import torch

from helpers import OracleBackend


def _vectors(dim, channel_last, device):
    if channel_last:
        shapes = [((1, 100, dim), (1, 100)), ((1, 8, 8, dim), (1, 8, 8)), ((1, 10, 8, 8, dim), (1, 10, 8, 8))]
    else:
        shapes = [((1, dim, 100), (1, 100)), ((1, dim, 8, 8), (1, 8, 8)), ((1, dim, 10, 8, 8), (1, 10, 8, 8))]
    return [(torch.randn(s, device=device), i) for s, i in shapes]


@pytest.fixture(params=["cpu-oracle", pytest.param("gpu-native", marks=pytest.mark.gpu)])
def device(request, oracle):
    from vector_quantization import search

    if request.param == "cpu-oracle":
        search.set_backend(OracleBackend)
        yield "cpu"
        search.set_backend(None)
    else:
        search.set_backend(None)
        yield "cuda:0"


def _vq(**kw):
    import vector_quantization as vq
    from vector_quantization.codebooks import CodebookParams

    dim = kw.pop("dim", 4)
    cdim = kw.get("codebook_dim", None) or dim
    params = CodebookParams(dim=cdim, codebook_size=kw.pop("codebook_size", 32), **kw.pop("cb", {}))
    return vq.VectorQuantize(dim=dim, codebook_params=params, **kw)


VQ_VARIANTS = {
    "default": dict(),
    "channel_first": dict(channel_last=False),
    "cosine": dict(cb=dict(use_cosine_sim=True)),
    "cosine_l2": dict(cb=dict(use_cosine_sim=True, transform_input="l2norm", weights_regularization="l2norm")),
    "multihead_separate": dict(dim=32, heads=4, codebook_dim=8, separate_codebook_per_head=True),
    "multihead_shared": dict(dim=32, heads=4, codebook_dim=8),
    "lower_codebook_dim": dict(dim=32, codebook_dim=16),
    "heads_with_projection": dict(dim=16, heads=2),
    "layernorm_projection": dict(dim=32, codebook_dim=16, layernorm_after_project_in=True),
}


@pytest.mark.parametrize("variant", sorted(VQ_VARIANTS))
def test_vector_quantize_shapes(device, variant):
    torch.manual_seed(0)
    kw = {k: (dict(v) if isinstance(v, dict) else v) for k, v in VQ_VARIANTS[variant].items()}
    dim = kw.get("dim", 4)
    heads = kw.get("heads", 1)
    mod = _vq(**kw).to(device)
    assert mod.training
    for feats, ishape in _vectors(dim, kw.get("channel_last", True), device):
        quantized, indices, loss = mod(feats)
        assert quantized.shape == feats.shape and quantized.dtype == torch.float32
        assert tuple(indices.shape) == (ishape + (heads,) if heads > 1 else ishape)
        assert indices.dtype == torch.int64
        assert loss.shape == (1,) and float(loss) >= 0.0
        assert int(indices.min()) >= 0 and int(indices.max()) < 32
    mod.eval()
    feats, _ = _vectors(dim, kw.get("channel_last", True), device)[0]
    q, i, loss = mod(feats)
    assert float(loss) == 0.0


def test_kmeans_variants(device):
    from vector_quantization.codebooks import KmeansParameters

    torch.manual_seed(0)
    for cb in (dict(initialization_by_kmeans=True, kmeans_params=KmeansParameters(iter=3)),
               dict(initialization_by_kmeans=True, kmeans_params=KmeansParameters(iter=3), use_cosine_sim=True)):
        mod = _vq(dim=8, codebook_size=16, cb=cb).to(device)
        x = torch.randn(2, 60, 8, device=device)
        q, i, loss = mod(x)
        assert q.shape == x.shape and i.shape == x.shape[:-1] and mod._codebook.is_initialized
    # fewer samples than codes
    mod = _vq(dim=8, codebook_size=64, cb=dict(initialization_by_kmeans=True,
                                               kmeans_params=KmeansParameters(iter=2))).to(device)
    q, i, loss = mod(torch.randn(1, 10, 8, device=device))
    assert q.shape == (1, 10, 8)


def test_two_dimensional_input_and_masks(device):
    torch.manual_seed(0)
    mod = _vq(dim=8).to(device)
    x = torch.randn(50, 8, device=device)
    q, i, loss = mod(x)
    assert q.shape == x.shape and i.shape == (50,)
    x = torch.randn(3, 20, 8, device=device)
    mask = torch.zeros(3, 20, dtype=torch.bool, device=device)
    mask[:, :7] = True
    q, i, loss = mod(x, mask=mask)
    assert torch.equal(q[~mask], x[~mask]) and loss.shape == (1,)


def test_loss_breakdown_and_codes_roundtrip(device):
    torch.manual_seed(0)
    mod = _vq(dim=8).to(device).eval()
    x = torch.randn(2, 30, 8, device=device)
    q, i, loss, breakdown = mod(x, return_loss_breakdown=True)
    assert breakdown._fields == ("commitment", "codebook_diversity", "orthogonal_reg", "inplace_optimize")
    assert torch.equal(mod.get_codes_from_indices(i), q)  # repaired w.r.t. the fork's broken `.embed`
    assert torch.equal(mod.get_output_from_indices(i), q)
    assert mod.codebook.shape == (32, 8)


RVQ_VARIANTS = {
    "default": dict(num_quantizers=4),
    "shared_codebook": dict(num_quantizers=4, shared_codebook=True),
    "projection": dict(num_quantizers=3, codebook_dim=8),
    "quantize_dropout": dict(num_quantizers=4, quantize_dropout=True, quantize_dropout_cutoff_index=1),
}


@pytest.mark.parametrize("variant", sorted(RVQ_VARIANTS))
def test_residual_vq_shapes(device, variant):
    import vector_quantization as vq
    from vector_quantization.codebooks import CodebookParams

    torch.manual_seed(0)
    kw = dict(RVQ_VARIANTS[variant])
    Q = kw["num_quantizers"]
    inner = kw.get("codebook_dim", None) or 16
    mod = vq.ResidualVQ(dim=16, codebook_params=CodebookParams(dim=inner, codebook_size=32), **kw).to(device)
    x = torch.randn(2, 40, 16, device=device)
    quantized, indices, losses = mod(x)
    assert quantized.shape == x.shape
    assert indices.shape == x.shape[:-1] + (Q,) and indices.dtype == torch.int64
    assert losses.shape == (1, Q)
    mod.eval()
    quantized, indices, losses, all_codes = mod(x, return_all_codes=True)
    assert all_codes.shape == (Q,) + x.shape[:-1] + (inner,)
    if variant != "projection":
        torch.testing.assert_close(all_codes.sum(0), quantized, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(mod.get_output_from_indices(indices), quantized, rtol=1e-5, atol=1e-5)


def test_grouped_residual_vq_shapes(device):
    import vector_quantization as vq
    from vector_quantization.codebooks import CodebookParams

    torch.manual_seed(0)
    mod = vq.GroupedResidualVQ(dim=32, groups=2, num_quantizers=3,
                               codebook_params=CodebookParams(dim=16, codebook_size=32)).to(device)
    x = torch.randn(2, 25, 32, device=device)
    for training in (True, False):
        mod.train(training)
        quantized, indices, losses = mod(x)
        assert quantized.shape == x.shape
        assert indices.shape == (2, 2, 25, 3)
        assert losses.shape == (2, 1, 3)
    q2, i2, l2, codes = mod(x, return_all_codes=True)
    assert codes.shape == (2, 3, 2, 25, 16)
    torch.testing.assert_close(mod.get_output_from_indices(i2), q2, rtol=1e-5, atol=1e-5)


def test_constructor_errors_match_reference():
    import vector_quantization as vq
    from vector_quantization.codebooks import Codebook, CodebookParams

    with pytest.raises(AssertionError):
        vq.ResidualVQ(dim=8, num_quantizers=2, heads=2, codebook_params=CodebookParams(dim=8, codebook_size=4))
    with pytest.raises(AssertionError):  # learnable codebook is incompatible with EMA (the default)
        vq.VectorQuantize(dim=8, codebook_params=CodebookParams(dim=8, codebook_size=4, learnable_codebook=True))
    with pytest.raises(TypeError):  # the reference raises a str -> TypeError
        Codebook(dim=8, codebook_size=4, transform_input="nope")
    with pytest.raises(AssertionError):
        vq.GroupedResidualVQ(dim=10, groups=3, num_quantizers=2, codebook_params=CodebookParams(dim=3, codebook_size=4))


def test_import_surface():
    import vector_quantization
    from vector_quantization import GroupedResidualVQ, ResidualVQ, VectorQuantize  # noqa: F401
    from vector_quantization.codebooks import (AffineParameters, Codebook, CodebookParams, GumbelParams,  # noqa: F401
                                               KmeansParameters)
    from vector_quantization.residual_vq import ResidualVQ as R2  # noqa: F401
    from vector_quantization.vector_quantize_pytorch import VectorQuantize as V2  # noqa: F401

    p = CodebookParams(dim=4, codebook_size=8)
    assert (p.decay, p.ema_update, p.threshold_ema_dead_code, p.use_cosine_sim) == (0.8, True, 2, False)
    assert vector_quantization.__all__


def test_random_projection_quantizer(device):
    """Repaired BEST-RQ quantizer: indices equal a manual LayerNorm -> projection -> cosine argmax."""
    import vector_quantization as vq

    torch.manual_seed(0)
    mod = vq.RandomProjectionQuantizer(dim=24, codebook_size=40, codebook_dim=8, num_codebooks=3).to(device)
    x = torch.randn(2, 30, 24, device=device)
    idx = mod(x)
    assert idx.shape == (2, 30, 3) and idx.dtype == torch.int64
    xn = torch.nn.functional.layer_norm(x, (24,))
    proj = torch.einsum("bnd,hde->bnhe", xn, mod.rand_projs)
    proj = torch.nn.functional.normalize(proj, dim=-1)
    codes = mod.vq._codebook.embeddings  # [h, K, e], l2-normalised at construction
    torch.testing.assert_close(codes.norm(dim=-1), torch.ones_like(codes[..., 0]), rtol=1e-5, atol=1e-5)
    sims = torch.einsum("bnhe,hke->bnhk", proj, codes)
    manual = sims.argmax(-1)
    agree = (manual == idx).float().mean().item()
    assert agree > 0.99, agree  # fp32 summation order may flip exact near-ties only
    picked = torch.gather(sims, -1, idx[..., None])[..., 0]
    torch.testing.assert_close(picked, sims.max(-1).values, rtol=0, atol=1e-5)
    # BEST-RQ training target: cross entropy of the (cosine) similarities against given labels
    labels = torch.randint(0, 40, (2, 30, 3), device=device)
    ce = mod(x, indices=labels)
    want = torch.nn.functional.cross_entropy(sims.permute(0, 3, 1, 2), labels)
    torch.testing.assert_close(ce, want, rtol=1e-5, atol=1e-5)


# ---------------------------------------------------------------------------------------------- stochastic sampling
def _stochastic_codebook(K, d, temperature, device):
    from vector_quantization.codebooks import Codebook, GumbelParams

    torch.manual_seed(1)
    cb = Codebook(dim=d, codebook_size=K, gumbel_params=GumbelParams(stochastic=True, temperature=temperature)).to(device)
    with torch.no_grad():
        cb.embeddings.copy_(torch.randn(1, K, d))
    return cb


def test_stochastic_sampling_limits_and_seed(device):
    """Gumbel-max code sampling (utils/general.py:106-129) is RNG-dependent -- no parity with the reference's draws --
    so it is pinned by its properties: temperature -> 0 is the deterministic argmax, equal seeds give equal draws."""
    from vector_quantization import search

    K, d = 64, 16
    x = torch.randn(1, 500, d, device=device)
    cold = _stochastic_codebook(K, d, 1e-6, device).eval()
    q, ind, _ = cold(x)
    det, best, _ = search.nearest_with_distance(x.reshape(1, -1, d), cold.embeddings)
    # similarities / 1e-6 are ~1e6 in fp32 (spacing 0.5), so the O(1) Gumbel noise can only decide between codes whose
    # distances agree to ~1e-5: the draw is the deterministic winner or tied with it to that precision
    picked = (x.reshape(-1, d) - cold.embeddings[0][ind.reshape(-1)]).norm(dim=-1)
    assert float((ind.reshape(-1) == det.reshape(-1)).float().mean()) > 0.98
    assert bool((picked <= best.reshape(-1) + 1e-4).all())
    assert torch.equal(q, cold.embeddings[0][ind])
    warm = _stochastic_codebook(K, d, 1.0, device).eval()  # eval: no EMA update between the calls; sampling stays on
    torch.manual_seed(7)
    a = warm(x)[1]
    torch.manual_seed(7)
    b = warm(x)[1]
    torch.manual_seed(8)
    c = warm(x)[1]
    assert torch.equal(a, b) and not torch.equal(a, c)


def test_stochastic_sampling_follows_the_softmax(device):
    """20000 copies of one row: code frequencies ~ softmax(similarities / temperature) (5 sigma)."""
    K, d, n, temp = 8, 4, 20000, 0.7
    cb = _stochastic_codebook(K, d, temp, device)
    row = torch.randn(1, 1, d, device=device) * 0.5
    x = row.expand(1, n, d).contiguous()
    torch.manual_seed(3)
    _, ind, sims = cb(x, return_similarities=True)
    prob = (sims[0, 0, 0] / temp).softmax(-1).double().cpu()
    freq = torch.bincount(ind.reshape(-1).cpu(), minlength=K).double() / n
    sigma = (prob * (1 - prob) / n).sqrt()
    assert bool(((freq - prob).abs() <= 5 * sigma + 1e-4).all()), (freq, prob)


def test_stochastic_in_vector_quantize_and_residual(device):
    import vector_quantization as vq
    from vector_quantization.codebooks import CodebookParams, GumbelParams

    params = CodebookParams(dim=8, codebook_size=32, gumbel_params=GumbelParams(stochastic=True))
    torch.manual_seed(0)
    mod = vq.VectorQuantize(dim=8, codebook_params=params).to(device).train()
    x = torch.randn(2, 50, 8, device=device, requires_grad=True)
    q, ind, loss = mod(x, freeze_codebook=True)
    assert q.shape == x.shape and ind.shape == (2, 50) and loss.shape == (1,)
    torch.testing.assert_close(q, mod._codebook.embeddings[0][ind])  # straight-through value == the sampled code
    (q.sum() + loss.sum()).backward()
    codes = mod._codebook.embeddings[0][ind]
    torch.testing.assert_close(x.grad, 1.0 + 2.0 * (x.detach() - codes) / x.numel(), rtol=1e-5, atol=1e-6)
    rvq = vq.ResidualVQ(dim=8, num_quantizers=3, shared_codebook=True, codebook_params=params).to(device)
    out, idx, losses, codes_all = rvq(x.detach(), return_all_codes=True)
    assert out.shape == x.shape and idx.shape == (2, 50, 3) and losses.shape == (1, 3) and codes_all.shape == (3, 2, 50, 8)
    with pytest.raises(NotImplementedError):
        vq.VectorQuantize(dim=8, codebook_params=CodebookParams(
            dim=8, codebook_size=32, gumbel_params=GumbelParams(stochastic=True, straight_through=True))).to(device)(x)


# ------------------------------------------------------------------------------- fused vs layer-by-layer training paths
@pytest.mark.parametrize("kind", ["rvq", "rvq_shared_frozen", "rvq_shared", "grvq"])
def test_fused_training_forward_equals_layer_by_layer(device, kind):
    """The fused launches (all stages / groups at once, native per-stage EMA statistics, per-group losses) must leave the
    same outputs, losses and updated codebooks as walking the layers one VectorQuantize at a time."""
    import copy

    import vector_quantization as vq
    from vector_quantization.codebooks import CodebookParams

    torch.manual_seed(0)
    params = CodebookParams(dim=32 if kind != "grvq" else 16, codebook_size=48, threshold_ema_dead_code=0)
    if kind == "grvq":
        fused = vq.GroupedResidualVQ(dim=32, groups=2, num_quantizers=3, codebook_params=params).to(device)
    else:
        fused = vq.ResidualVQ(dim=32, num_quantizers=3, shared_codebook=kind.startswith("rvq_shared"),
                              codebook_params=params).to(device)
    plain = copy.deepcopy(fused)
    rvqs = plain.rvqs if kind == "grvq" else [plain]
    for m in [plain] + list(rvqs):
        m._fusable = lambda *a, **k: False  # layer-by-layer
    fused.train()
    plain.train()
    x = torch.randn(4, 200, 32, device=device)
    w = torch.randn(4, 200, 32, device=device)
    kw = dict(freeze_codebook=True) if kind == "rvq_shared_frozen" else {}
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    a = fused(xa, **kw)
    b = plain(xb, **kw)
    assert torch.equal(a[1], b[1])
    torch.testing.assert_close(a[0], b[0], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(a[2], b[2], rtol=1e-5, atol=1e-6)
    # gradients: straight-through identity per stage + commitment losses (found missing on the grouped fused path once)
    ((a[0] * w).sum() + a[2].sum() * 2.0).backward()
    ((b[0] * w).sum() + b[2].sum() * 2.0).backward()
    torch.testing.assert_close(xa.grad, xb.grad, rtol=1e-5, atol=1e-6)
    # (a shared codebook under EMA is searched stage by stage on both sides: the module must not fuse that case)
    for ma, mb in zip(fused.modules(), plain.modules()):
        if isinstance(ma, vq.Codebook):
            torch.testing.assert_close(ma.embeddings, mb.embeddings, rtol=1e-4, atol=1e-5)
            torch.testing.assert_close(ma.cluster_size, mb.cluster_size, rtol=1e-6, atol=1e-6)
