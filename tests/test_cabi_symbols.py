"""The C-ABI library loads and exports every symbol include/vq_mi355x.h declares (no compute without a GPU)."""
from __future__ import annotations

import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "vq_mi355x.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vq_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported():
    from vector_quantization import native

    if not os.path.exists(native.lib_path()):
        import __graft_entry__

        __graft_entry__.build()
    lib = ctypes.CDLL(native.lib_path())
    names = _declared()
    assert len(names) >= 10
    for n in names:
        assert hasattr(lib, n), f"{n} declared in vq_mi355x.h but not exported"
    assert set(native.EXPORTED_SYMBOLS) == set(names)


def test_argument_validation_without_gpu():
    from vector_quantization import native

    lib = native.load()
    assert lib.vq_packed_floats(0, 4) == 0
    assert lib.vq_packed_floats(1024, 256) == 1024 * 260 + 2048
    assert lib.vq_packed_floats(33, 100) == 64 * 132 + 2048
    assert lib.vq_workspace_bytes(1, 1024, 1) > 1024 * 8
    a = native.VqArgs()
    rc = lib.vq_quantize_f32(ctypes.byref(a), None)
    assert rc == -1 and b"non-positive" in lib.vq_last_error()
    assert lib.vq_quantize_f32(None, None) == -1


def test_cpu_tensors_fail_loudly():
    """No silent CPU fallback: the product path raises when handed CPU tensors."""
    import torch
    import vector_quantization as vq
    from vector_quantization import native
    from vector_quantization.codebooks import CodebookParams

    m = vq.VectorQuantize(dim=16, codebook_params=CodebookParams(dim=16, codebook_size=32)).eval()
    with pytest.raises(native.NativeUnavailable):
        m(torch.randn(2, 5, 16))
    r = vq.ResidualVQ(dim=16, num_quantizers=2, codebook_params=CodebookParams(dim=16, codebook_size=32)).eval()
    with pytest.raises(native.NativeUnavailable):
        r(torch.randn(2, 5, 16))


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from vector_quantization import native

    monkeypatch.setattr(native, "_lib", None)
    monkeypatch.setattr(native, "_LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(native.NativeUnavailable):
        native.load()
