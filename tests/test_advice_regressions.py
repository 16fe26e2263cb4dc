"""CPU regressions for the round-2 advisor findings (host logic, checker backend -- no GPU)."""
from __future__ import annotations

import pytest
import torch

from helpers import OracleBackend


@pytest.fixture(autouse=True)
def _oracle_backend(oracle):
    from vector_quantization import search

    search.set_backend(OracleBackend)
    yield
    search.set_backend(None)


def _modules():
    import vector_quantization as vq
    from vector_quantization.codebooks import CodebookParams

    yield vq.VectorQuantize(dim=16, codebook_params=CodebookParams(dim=16, codebook_size=32))
    yield vq.ResidualVQ(dim=16, num_quantizers=3, codebook_params=CodebookParams(dim=16, codebook_size=32))
    yield vq.GroupedResidualVQ(dim=16, groups=2, num_quantizers=2, codebook_params=CodebookParams(dim=8, codebook_size=16))


def test_eval_forward_under_inference_mode_with_fresh_modules():
    """quantizer._cached_zeros read ``_version`` of a tensor created under inference mode (an inference tensor has none)."""
    torch.manual_seed(0)
    x = torch.randn(2, 9, 16)
    for mod in _modules():
        mod.eval()
        with torch.inference_mode():
            q1, i1, l1 = mod(x)
            q2, i2, l2 = mod(x)
        with torch.no_grad():
            q3, i3, l3 = mod(x)
        assert torch.equal(i1, i3) and torch.equal(q1, q3) and torch.equal(i1, i2)
        assert float(l1.sum()) == 0.0 and l1.shape == l3.shape


def test_modules_built_under_inference_mode_still_run():
    """embeddings that ARE inference tensors have no version counter: the packed-image cache key must not read it."""
    torch.manual_seed(0)
    x = torch.randn(2, 9, 16)
    with torch.inference_mode():
        mods = [m.eval() for m in _modules()]
        for mod in mods:
            q, i, _ = mod(x)
            assert q.shape == x.shape


def test_zero_loss_is_recreated_after_an_in_place_edit():
    import vector_quantization as vq
    from vector_quantization.codebooks import CodebookParams

    mod = vq.VectorQuantize(dim=16, codebook_params=CodebookParams(dim=16, codebook_size=32)).eval()
    x = torch.randn(2, 9, 16)
    with torch.no_grad():
        _, _, l1 = mod(x)
        l1 += 3.0
        _, _, l2 = mod(x)
    assert float(l2) == 0.0


def test_deterministic_mode_refuses_rows_wider_than_2048_dims():
    from vector_quantization import search

    was = torch.are_deterministic_algorithms_enabled()
    warn_only = torch.is_deterministic_algorithms_warn_only_enabled()
    try:
        torch.use_deterministic_algorithms(True)
        assert search._want_deterministic(2048) is True
        with pytest.raises(RuntimeError, match="2048"):
            search._want_deterministic(4096)
        torch.use_deterministic_algorithms(True, warn_only=True)
        with pytest.warns(UserWarning, match="2048"):
            assert search._want_deterministic(4096) is False
    finally:
        torch.use_deterministic_algorithms(was, warn_only=warn_only)
    assert search._want_deterministic(4096) is False or was


def test_packed_image_of_another_shape_is_refused():
    from vector_quantization import native

    try:
        native.load()
    except native.NativeUnavailable:
        pytest.skip("library not built")
    good = native.packed_floats(32, 16)
    packed = torch.zeros((1, good), dtype=torch.float32)
    native._check_packed(packed, 1, 32, 16, packed.device)
    with pytest.raises(ValueError, match="does not belong"):
        native._check_packed(packed, 1, 300, 16, packed.device)
    with pytest.raises(ValueError, match="does not belong"):
        native._check_packed(packed.double(), 1, 32, 16, packed.device)
    with pytest.raises(ValueError, match="does not belong"):
        native._check_packed(torch.zeros((2, good)), 1, 32, 16, packed.device)
