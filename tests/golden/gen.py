"""Seeded synthetic inputs shared by make_golden.py (which imports the reference) and the tests
(which never do).  Everything is regenerated from seeds; fixtures carry checksums of the regenerated
tensors so a drift in the RNG stream is detected instead of silently changing the test.

Parity classes:
  S     separated       x ~ randn, codebook ~ randn                       -> indices must be 100 % equal
  G     exact grid      x, codebook in {-16..16}/8 (all fp32 sums exact)   -> 100 % equal incl. sqrt ties
  Gdup  exact grid with the second half of the codebook duplicating the first (forces ties;
        every index must come from the first half)
  R     reference-default init: codebook ~ U(+-sqrt(6/(K*D))) (kaiming_uniform_ on [h,K,D],
        /root/reference/vector_quantization/utils/general.py:101-104) -> near-tie heavy
"""
from __future__ import annotations

import math

import torch

X_SEED = 1234
CB_SEED = 4321


def _gen(seed: int) -> torch.Generator:
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    return g


def make_x(shape, cls: str = "S", seed: int = X_SEED) -> torch.Tensor:
    g = _gen(seed)
    if cls in ("G", "Gdup"):
        return torch.randint(-16, 17, tuple(shape), generator=g).float() / 8.0
    return torch.randn(tuple(shape), generator=g)


def make_codebook(h: int, K: int, D: int, cls: str = "S", seed: int = CB_SEED, scale: float = 1.0) -> torch.Tensor:
    g = _gen(seed)
    if cls == "G":
        cb = torch.randint(-16, 17, (h, K, D), generator=g).float() / 8.0
    elif cls == "Gdup":
        half = torch.randint(-16, 17, (h, K // 2, D), generator=g).float() / 8.0
        cb = torch.cat([half, half], dim=1)
        if cb.shape[1] < K:  # odd K: pad with a far-away row
            cb = torch.cat([cb, torch.full((h, K - cb.shape[1], D), 64.0)], dim=1)
    elif cls == "R":
        bound = math.sqrt(6.0 / (K * D))
        cb = (torch.rand((h, K, D), generator=g) * 2.0 - 1.0) * bound
    else:
        cb = torch.randn((h, K, D), generator=g)
    return cb * scale


def make_rvq_codebooks(Q: int, K: int, D: int, cls: str = "S", seed: int = CB_SEED) -> torch.Tensor:
    """[Q, K, D]; stage i scaled by 2^(-i/2) so residuals stay on scale."""
    cbs = []
    for i in range(Q):
        if cls == "S":
            cbs.append(make_codebook(1, K, D, "S", seed + i, scale=2.0 ** (-i / 2.0))[0])
        else:
            cbs.append(make_codebook(1, K, D, cls, seed + i)[0])
    return torch.stack(cbs)


def checksum(t: torch.Tensor):
    t64 = t.detach().double().flatten()
    return [float(t64.sum()), float(t64.abs().sum()), float(t64[0]), float(t64[-1])]


def l2norm(t: torch.Tensor) -> torch.Tensor:
    return torch.nn.functional.normalize(t, p=2, dim=-1)


def seeded_projection_(mod, seed: int = 5150) -> None:
    """Overwrite the Linear projections of a VectorQuantize-like module (``project_in`` / ``project_out``, reference and
    drop-in alike) with seeded weights, so that large projections need not be stored in a fixture."""
    g = _gen(seed)
    lin_in = mod.project_in if isinstance(mod.project_in, torch.nn.Linear) else mod.project_in[0]
    with torch.no_grad():
        for lin in (lin_in, mod.project_out):
            fan_in = lin.weight.shape[1]
            lin.weight.copy_(torch.randn(tuple(lin.weight.shape), generator=g) / math.sqrt(fan_in))
            lin.bias.copy_(torch.randn(tuple(lin.bias.shape), generator=g) * 0.1)


_VALUES = {"nan": float("nan"), "inf": float("inf"), "-inf": float("-inf")}


def poison_(x: torch.Tensor, cb: torch.Tensor, spec) -> None:
    """Non-finite entries written in place (the ``nonfinite`` field of a case): ``spec["x"]`` = [[row, dim | None, value]]
    on the rows of ``x`` flattened to [-1, last dim] (dim None: the whole row), ``spec["cb"]`` = [[codebook, code, dim, value]]
    on ``cb`` [h | Q, K, D]; value in "nan", "inf", "-inf"."""
    if not spec:
        return
    rows = x.view(-1, x.shape[-1])
    for row, dim, val in spec.get("x", []):
        if dim is None:
            rows[row, :] = _VALUES[val]
        else:
            rows[row, dim] = _VALUES[val]
    for book, code, dim, val in spec.get("cb", []):
        cb[book, code, dim] = _VALUES[val]
