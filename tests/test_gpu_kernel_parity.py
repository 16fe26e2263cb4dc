"""GPU parity proper: HIP kernels (through the C ABI) vs the CPU oracle, bit-exact.

Indices AND winning distances must be identical (the MFMA chain order is the oracle's chain order);
quantized rows are exact gathers; squared-error sums agree to 1e-6 relative.
"""
from __future__ import annotations

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from gen import make_codebook, make_rvq_codebooks, make_x  # noqa: E402


def _native():
    from vector_quantization import native

    native.load()
    return native


def _run(native, x, cb, metric, **kw):
    dev = torch.device("cuda:0")
    r = native.quantize(x.to(dev), cb.to(dev), metric=metric, **kw)
    torch.cuda.synchronize()
    return {k: (v.cpu() if v is not None else None) for k, v in r.items()}


SHAPES = [
    # (H, M, K, D)
    (1, 8192, 256, 64),
    (1, 4096, 1024, 256),
    (1, 1024, 8192, 256),
    (8, 256, 8192, 64),
    (1, 111, 300, 100),
    (2, 33, 7, 5),
    (1, 40, 1, 16),
    (1, 257, 1000, 128),
    (1, 64, 4096, 512),
    (1, 1000, 96, 32),
    (1, 130, 70, 24),
]


@pytest.mark.parametrize("H,M,K,D", SHAPES)
@pytest.mark.parametrize("cls", ["S", "G", "Gdup", "R"])
@pytest.mark.parametrize("metric", [0, 1])
def test_single_stage_bit_exact(oracle, H, M, K, D, cls, metric):
    native = _native()
    x = make_x((H, M, D), cls)
    cb = make_codebook(H, K, D, cls)
    ref = oracle.vq_forward(x.numpy(), cb.numpy(), metric, training=False)
    got = _run(native, x, cb[:, None].contiguous(), metric, want_sq_err=True)
    idx = got["idx"][..., 0].numpy()
    np.testing.assert_array_equal(idx, ref["idx"])
    assert np.array_equal(got["best"][..., 0].numpy().view(np.uint32), ref["best"].view(np.uint32)), "distances differ"
    np.testing.assert_array_equal(got["out"].numpy(), ref["out"])
    np.testing.assert_allclose(got["sq_err"].numpy()[0], ref["sq_err"], rtol=1e-6)
    if cls == "Gdup" and K >= 2 and K % 2 == 0:
        assert idx.max() < max(K // 2, 1)


@pytest.mark.parametrize("flags_name", ["F_FORCE_SIMPLE", "F_FORCE_SPLIT"])
@pytest.mark.parametrize("H,M,K,D", [(1, 300, 1000, 256), (2, 100, 333, 48), (1, 64, 4096, 512)])
@pytest.mark.parametrize("metric", [0, 1])
def test_alternate_paths_bit_exact(oracle, flags_name, H, M, K, D, metric):
    native = _native()
    x = make_x((H, M, D), "S")
    cb = make_codebook(H, K, D, "S")
    ref = oracle.vq_forward(x.numpy(), cb.numpy(), metric, training=True)
    got = _run(native, x, cb[:, None].contiguous(), metric, want_sq_err=True, ste=True,
               flags=getattr(native, flags_name))
    np.testing.assert_array_equal(got["idx"][..., 0].numpy(), ref["idx"])
    assert np.array_equal(got["best"][..., 0].numpy().view(np.uint32), ref["best"].view(np.uint32))
    np.testing.assert_array_equal(got["out"].numpy(), ref["out"])
    np.testing.assert_allclose(got["sq_err"].numpy()[0], ref["sq_err"], rtol=1e-6)


@pytest.mark.parametrize("Q,M,K,D", [(8, 1024, 1024, 256), (4, 300, 256, 64), (3, 77, 100, 40), (2, 64, 512, 512),
                                     (5, 200, 64, 128)])
@pytest.mark.parametrize("training", [False, True])
@pytest.mark.parametrize("cls", ["S", "G"])
def test_residual_bit_exact(oracle, Q, M, K, D, training, cls, residual_plan):
    native = _native()
    x = make_x((M, D), cls)
    cbs = make_rvq_codebooks(Q, K, D, cls)
    ref = oracle.rvq_forward(x.numpy(), cbs.numpy(), 0, training=training)
    got = _run(native, x[None], cbs[None].contiguous(), 0, ste=training, want_sq_err=True)
    np.testing.assert_array_equal(got["idx"][0].numpy(), ref["idx"])
    assert np.array_equal(got["best"][0].numpy().view(np.uint32), ref["best"].view(np.uint32))
    np.testing.assert_array_equal(got["out"][0].numpy(), ref["out"])
    np.testing.assert_allclose(got["sq_err"].numpy(), ref["sq_err"], rtol=1e-6)


def test_strided_rows_and_heads(oracle):
    """Head-split view 'b n (h d) -> h (b n) d' is searched in place (no copy)."""
    native = _native()
    H, M, D, K = 4, 200, 64, 512
    x = make_x((M, H * D), "S")
    cb = make_codebook(H, K, D, "S")
    xv = x.view(M, H, D).permute(1, 0, 2)  # [H, M, D], row stride H*D, head stride D
    ref = oracle.vq_forward(np.ascontiguousarray(xv.numpy()), cb.numpy(), 0)
    dev = torch.device("cuda:0")
    xg = x.to(dev)
    out = torch.empty_like(xg)
    r = native.quantize(xg.view(M, H, D).permute(1, 0, 2), cb[:, None].contiguous().to(dev), metric=0,
                        out=out.view(M, H, D).permute(1, 0, 2))
    torch.cuda.synchronize()
    np.testing.assert_array_equal(r["idx"][..., 0].cpu().numpy(), ref["idx"])
    np.testing.assert_array_equal(out.cpu().view(M, H, D).permute(1, 0, 2).numpy(), ref["out"])
