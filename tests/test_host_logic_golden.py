"""CPU suite: the drop-in modules' HOST logic (layouts, heads, projections, masks, losses, residual loop)
against the golden vectors captured from the imported reference, with the CPU oracle standing in for the
native op (tests-only checker backend).  This also pins the oracle itself to the reference: indices must be
identical on the S / G / Gdup classes, quantized vectors and losses within 1e-5.
"""
from __future__ import annotations

import pytest

from cases import CASES, LOSS_CASES
from build_case import build
from check_case import compare, compare_loss
from helpers import OracleBackend, checksum_close, load_golden


@pytest.fixture(autouse=True)
def _oracle_backend(oracle):
    from vector_quantization import search

    search.set_backend(OracleBackend)
    yield
    search.set_backend(None)


SLOW = {"cfg5_S"}


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_module_matches_reference_golden(case):
    arrays, meta = load_golden(case["name"])
    mod, x, kwargs, cb = build(case, arrays)
    ok, got = checksum_close(x, meta["x_checksum"])
    assert ok, f"input regeneration drifted: {got} vs {meta['x_checksum']}"
    ok, got = checksum_close(cb, meta["cb_checksum"])
    assert ok, f"codebook regeneration drifted: {got} vs {meta['cb_checksum']}"
    if "forward_seed" in case:
        import torch

        torch.manual_seed(case["forward_seed"])
    outputs = mod(x, **kwargs)
    compare(case, arrays, meta, outputs, x, cb, mod)


@pytest.mark.parametrize("case", LOSS_CASES, ids=[c["name"] for c in LOSS_CASES])
def test_similarity_consuming_losses_match_reference_golden(case):
    """Cross-entropy commitment, cross entropy to given indices, diversity loss: value, outputs and d loss / d x."""
    arrays, meta = load_golden(case["name"])
    mod, x, kwargs, cb = build(case, arrays)
    ok, got = checksum_close(x, meta["x_checksum"])
    assert ok, f"input regeneration drifted: {got} vs {meta['x_checksum']}"
    compare_loss(case, arrays, meta, mod, x, kwargs)


def test_orthogonal_loss_matches_reference_function():
    import numpy as np
    import torch

    from gen import CB_SEED, make_codebook
    from vector_quantization.losses import orthogonal_loss

    arrays, _ = load_golden("orthogonal_fn")
    for i in range(3):
        h, k, d = (int(v) for v in arrays[f"shape{i}"])
        cb = make_codebook(h, k, d, "S", seed=CB_SEED + i).requires_grad_(True)
        v = orthogonal_loss(cb)
        v.backward()
        np.testing.assert_allclose(float(v), float(arrays[f"value{i}"]), rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(cb.grad[:, :4].numpy(), arrays[f"grad_rows{i}"], rtol=1e-4, atol=1e-9)


@pytest.mark.parametrize("name", ["ce_commit", "ce_commit_mh_sep", "ce_indices_ignore", "div_t1", "div_mh_shared", "div_ce"])
def test_losses_are_chunk_size_independent(name, monkeypatch):
    """Force the smallest row chunks (128 rows): the multi-chunk forward / backward must still match the golden."""
    from cases import LOSS_CASES_BY_NAME
    from vector_quantization import losses

    monkeypatch.setattr(losses, "CHUNK_BYTES", 1)
    case = LOSS_CASES_BY_NAME[name]
    arrays, meta = load_golden(name)
    mod, x, kwargs, _ = build(case, arrays)
    assert x.numel() // x.shape[-1] > 128 or case.get("heads", 1) > 1
    compare_loss(case, arrays, meta, mod, x, kwargs)


def test_similarities_third_return_value():
    """Codebook.forward(return_similarities=True) gives the reference's third return value, with gradients."""
    import torch

    from vector_quantization.codebooks import Codebook

    torch.manual_seed(0)
    cb = Codebook(dim=16, codebook_size=32)
    cb.eval()
    x = torch.randn(2, 9, 16, requires_grad=True)
    q, ind, sims = cb(x, return_similarities=True)
    assert sims.shape == (1, 2, 9, 32)
    want = -torch.cdist(x.detach()[None].reshape(1, 18, 16), cb.embeddings).reshape(1, 2, 9, 32)
    torch.testing.assert_close(sims, want, rtol=1e-5, atol=1e-5)
    assert torch.equal(sims.argmax(-1)[0], ind)
    w = torch.randn_like(sims)
    (sims * w).sum().backward()
    x2 = x.detach().clone().requires_grad_(True)
    ((-torch.cdist(x2[None].reshape(1, 18, 16), cb.embeddings)).reshape(1, 2, 9, 32) * w).sum().backward()
    torch.testing.assert_close(x.grad, x2.grad, rtol=1e-4, atol=1e-5)
