"""GPU sweep: the same configuration lists as the container-only reference sweep (tests/golden/sweep_against_reference.py,
which pins the CPU checker-backend modules to the imported reference), here comparing the NATIVE modules on the device
with the checker-backend modules on the CPU: outputs, losses, updated training state, and all gradients.  Together the two
sweeps tie the HIP kernels to the reference over ~250 module configurations without the reference travelling."""
from __future__ import annotations

import copy

import pytest
import torch

pytestmark = pytest.mark.gpu

from helpers import OracleBackend  # noqa: E402
from sweep_configs import autograd_cases, forward_cases, materialise_forward_kwargs  # noqa: E402

DEV = "cuda:0"


def _close(name, a, b, tol=1e-5):
    if isinstance(a, (tuple, list)):
        assert len(a) == len(b), name
        for i, (u, v) in enumerate(zip(a, b)):
            _close(f"{name}[{i}]", u, v, tol)
        return
    a, b = torch.as_tensor(a).detach().cpu(), torch.as_tensor(b).detach().cpu()
    assert a.shape == b.shape, f"{name}: {tuple(a.shape)} vs {tuple(b.shape)}"
    if a.dtype in (torch.int64, torch.int32):
        assert torch.equal(a, b), f"{name}: {int((a != b).sum())} of {a.numel()} indices differ"
    else:
        scale = max(1.0, float(b.abs().max())) if b.numel() else 1.0
        err = float((a.double() - b.double()).abs().max()) if a.numel() else 0.0
        assert err <= tol * scale, f"{name}: max abs err {err}"


def _build(kind, ctor):
    import vector_quantization as vq
    from vector_quantization.codebooks import CodebookParams

    ctor = {k: (dict(v) if isinstance(v, dict) else v) for k, v in ctor.items()}
    cb_kw = ctor.pop("cb")
    torch.manual_seed(7)
    return getattr(vq, kind)(codebook_params=CodebookParams(**cb_kw), **ctor), ctor, cb_kw


def _to_dev(kw):
    return {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in kw.items()}


FWD = [(i, c) for i, c in enumerate(forward_cases())]


@pytest.mark.parametrize("mode", ["eval", "train_frozen", "train_ema"])
def test_native_modules_equal_checker_backend_forward(mode, oracle):
    from vector_quantization import search

    failures = []
    for i, (kind, ctor, shape, fwd) in FWD:
        cpu_mod, ctor_kw, cb_kw = _build(kind, ctor)
        gpu_mod = copy.deepcopy(cpu_mod).to(DEV)
        x = torch.randn(*shape, generator=torch.Generator().manual_seed(11))
        kw = materialise_forward_kwargs(ctor_kw, cb_kw, x, fwd)
        if mode == "train_frozen":
            kw["freeze_codebook"] = True
        for m in (cpu_mod, gpu_mod):
            m.train() if mode != "eval" else m.eval()
        try:
            search.set_backend(OracleBackend)
            try:
                with torch.no_grad():
                    want = cpu_mod(x, **kw)
            except Exception as e_cpu:  # noqa: BLE001  (e.g. the fork's own shape bug with image input + quantize dropout)
                search.set_backend(None)
                with pytest.raises(type(e_cpu)), torch.no_grad():
                    gpu_mod(x.to(DEV), **_to_dev(kw))
                continue
            finally:
                search.set_backend(None)
            with torch.no_grad():
                got = gpu_mod(x.to(DEV), **_to_dev(kw))
            _close("out", got, want)
            for (k, a), (_, b) in zip(gpu_mod.state_dict().items(), cpu_mod.state_dict().items()):
                _close(f"state[{k}]", a, b, tol=1e-4)
        except AssertionError as e:
            failures.append(f"#{i} {kind} {ctor} {fwd} {mode}: {e}")
    assert not failures, "\n".join(failures[:10])


def test_native_modules_equal_checker_backend_gradients(oracle):
    from vector_quantization import search

    failures = []
    for i, (kind, ctor, shape, fwd) in enumerate(autograd_cases()):
        for mode in ("train_frozen", "train_ema"):
            cpu_mod, ctor_kw, cb_kw = _build(kind, ctor)
            gpu_mod = copy.deepcopy(cpu_mod).to(DEV)
            x = torch.randn(*shape, generator=torch.Generator().manual_seed(11))
            kw = materialise_forward_kwargs(ctor_kw, cb_kw, x, fwd)
            if mode == "train_frozen":
                kw["freeze_codebook"] = True
            grads = []
            try:
                for m, dev, backend in ((cpu_mod, "cpu", OracleBackend), (gpu_mod, DEV, None)):
                    m.train()
                    search.set_backend(backend)
                    try:
                        xs = x.to(dev).clone().requires_grad_(True)
                        out = m(xs, **(_to_dev(kw) if dev != "cpu" else kw))
                        w = torch.randn(out[0].shape, generator=torch.Generator().manual_seed(5)).to(dev)
                        ((out[0] * w).sum() + out[2].sum() * 1.5).backward()
                        grads.append((xs.grad, {k: p.grad for k, p in m.named_parameters()}))
                    finally:
                        search.set_backend(None)
                _close("x.grad", grads[1][0], grads[0][0])
                for k, g in grads[0][1].items():
                    if g is not None or grads[1][1][k] is not None:
                        _close(f"grad[{k}]", grads[1][1][k], g, tol=1e-4)
            except AssertionError as e:
                failures.append(f"#{i} {kind} {ctor} {fwd} {mode}: {e}")
    assert not failures, "\n".join(failures[:10])


def _scaled(ctor, shape):
    """The same configuration at realistic sizes: dim 32 -> 256, codebook dims x8, 40 codes -> 1000, 60 rows -> 4096
    (other kernel instantiations: Dp = 128 / 256, multi-tile sweeps, tail masking at K % 32 != 0)."""
    c = {k: (dict(v) if isinstance(v, dict) else v) for k, v in ctor.items()}
    if c.get("dim") == 32:
        c["dim"] = 256
    if c.get("codebook_dim"):
        c["codebook_dim"] *= 8
    cb = c["cb"]
    cb["dim"] *= 8
    cb["codebook_size"] = 1000
    if len(shape) == 3:
        shape = (8, 512, 256)
    elif len(shape) == 2:
        shape = (4096, 256)
    elif shape[1] == 32:  # channel-first image
        shape = (8, 256, 16, 32)
    else:
        shape = (8, 16, 32, 256)
    return c, shape


@pytest.mark.parametrize("mode", ["eval", "train_ema"])
def test_native_modules_equal_checker_backend_at_realistic_sizes(mode, oracle):
    from vector_quantization import search

    failures = []
    picked = [c for c in forward_cases() if "given_indices" not in c[3] and not c[1].get("quantize_dropout")
              and c[1].get("dim") == 32][::5]  # (the dim-32 configurations are the ones _scaled knows how to enlarge)
    n_run = 0
    sized = [(kind, *_scaled(ctor, shape), fwd) for kind, ctor, shape, fwd in picked]
    # rows wider than 512 dims at realistic sizes (sliced sweep: several slices, K % 32 != 0, EMA over wide rows)
    noexp = dict(threshold_ema_dead_code=0)
    sized.append(("VectorQuantize", dict(dim=768, cb=dict(dim=768, codebook_size=1000, **noexp)), (8, 512, 768), {}))
    sized.append(("ResidualVQ", dict(dim=640, num_quantizers=3, cb=dict(dim=640, codebook_size=500, **noexp)), (4, 256, 640), {}))
    for i, (kind, ctor, shape, fwd) in enumerate(sized):
        cpu_mod, ctor_kw, cb_kw = _build(kind, ctor)
        if getattr(cpu_mod, "has_projections", False):
            continue  # nn.Linear in front: MKL vs rocBLAS rounding moves near-ties at these sizes (not this library's kernels)
        n_run += 1
        gpu_mod = copy.deepcopy(cpu_mod).to(DEV)
        x = torch.randn(*shape, generator=torch.Generator().manual_seed(11))
        kw = materialise_forward_kwargs(ctor_kw, cb_kw, x, fwd)
        grads = []
        try:
            for m, dev, backend in ((cpu_mod, "cpu", OracleBackend), (gpu_mod, DEV, None)):
                m.train() if mode != "eval" else m.eval()
                search.set_backend(backend)
                try:
                    xs = x.to(dev).clone().requires_grad_(mode != "eval")
                    out = m(xs, **(_to_dev(kw) if dev != "cpu" else kw))
                    if mode != "eval":
                        w = torch.randn(out[0].shape, generator=torch.Generator().manual_seed(5)).to(dev)
                        ((out[0] * w).sum() + out[2].sum() * 1.5).backward()
                    grads.append((out, xs.grad))
                finally:
                    search.set_backend(None)
            _close("out", grads[1][0], grads[0][0], tol=2e-5)
            if mode != "eval":
                _close("x.grad", grads[1][1], grads[0][1], tol=2e-5)
                for (k, a), (_, b) in zip(gpu_mod.state_dict().items(), cpu_mod.state_dict().items()):
                    _close(f"state[{k}]", a, b, tol=1e-4)
        except AssertionError as e:
            failures.append(f"#{i} {kind} {ctor} {shape} {fwd} {mode}: {e}")
    assert n_run >= 8, n_run
    assert not failures, "\n".join(failures[:10])
