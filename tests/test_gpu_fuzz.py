"""GPU: a bounded run of tools/fuzz_kernels.py inside the suite (VERDICT r2 #7d), fixed seed -- random shapes over every launch
path (one-block, persistent, wave-pair, K split, main + tail, wide rows, residual stacks), exact-grid ties and non-finite entries,
against the scalar kernel / the CPU oracle, bit for bit."""
from __future__ import annotations

import os
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_fuzz_kernels_fixed_seed():
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_kernels

    counts = fuzz_kernels.run(budget=float(os.environ.get("VQ_FUZZ_SECONDS", "75")), seed=20261005, verbose=False)
    assert counts["single"] >= 20 and counts["residual"] >= 10 and counts["poisoned"] >= 3, counts
