"""CPU suite: the drop-in modules' HOST logic (layouts, heads, projections, masks, losses, residual loop)
against the golden vectors captured from the imported reference, with the CPU oracle standing in for the
native op (tests-only checker backend).  This also pins the oracle itself to the reference: indices must be
identical on the S / G / Gdup classes, quantized vectors and losses within 1e-5.
"""
from __future__ import annotations

import pytest

from cases import CASES
from build_case import build
from check_case import compare
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
    outputs = mod(x, **kwargs)
    compare(case, arrays, meta, outputs, x, cb, mod)
