"""GPU: the modules' cached packed images / stacked stage codebooks and the launch budget of an
inference forward.

* an inference forward of VectorQuantize / ResidualVQ enqueues exactly ONE kernel once the cache is warm
  (counted with torch.profiler's device activity; a kernel trace of the same forward is kept under profiles/);
* every way the code values can change is seen by the next forward: in-place ``copy_`` / ``load_state_dict``
  (version counter), the native EMA step and dead-code re-seeding (explicit invalidation), ``.to(device)``;
* stacks longer than one launch's LDS budget (vq_max_fused_stages) run layer by layer and still match the oracle.
"""
from __future__ import annotations

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _vq(dim=64, K=256, **kw):
    import vector_quantization as vq
    from vector_quantization.codebooks import CodebookParams

    mod = vq.VectorQuantize(dim=dim, codebook_params=CodebookParams(dim=dim, codebook_size=K, **kw))
    with torch.no_grad():
        mod._codebook.embeddings.copy_(torch.randn(mod._codebook.embeddings.shape, generator=torch.Generator().manual_seed(5)))
    return mod.to(DEV)


def _rvq(dim=64, K=128, Q=4, **kw):
    import vector_quantization as vq
    from vector_quantization.codebooks import CodebookParams

    mod = vq.ResidualVQ(dim=dim, num_quantizers=Q, codebook_params=CodebookParams(dim=dim, codebook_size=K, **kw))
    g = torch.Generator().manual_seed(6)
    with torch.no_grad():
        for i, layer in enumerate(mod.layers):
            layer._codebook.embeddings.copy_(torch.randn((1, K, dim), generator=g) * 2.0 ** (-i / 2.0))
    return mod.to(DEV)


def _kernel_names(fn):
    """Names of the device kernels one call of fn() enqueues (torch.profiler, device activity)."""
    from torch.profiler import ProfilerActivity, profile

    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        fn()
        torch.cuda.synchronize()
    names = []
    for ev in prof.events():
        if str(getattr(ev, "device_type", "")).endswith("CUDA") and ev.name and not ev.name.lower().startswith("memcpy") \
                and not ev.name.lower().startswith("memset"):
            names.append(ev.name)
    return names


@pytest.mark.parametrize("kind", ["vq", "rvq"])
def test_inference_forward_is_one_kernel(kind):
    mod = (_vq() if kind == "vq" else _rvq()).eval()
    # enough rows for the fused launch (a handful of rows is searched split-K: keys + finalize kernels, by design)
    x = torch.randn((64, 1024, 64), generator=torch.Generator().manual_seed(1)).to(DEV)
    with torch.no_grad():
        mod(x)  # warms the cache (this call packs)
        names = _kernel_names(lambda: mod(x))
    if not names:
        pytest.skip("torch.profiler reported no device activity on this build")
    assert len(names) == 1 and "vq_search_mfma" in names[0], names


def test_inplace_weight_load_is_seen(oracle):
    mod = _vq().eval()
    x = torch.randn((2, 200, 64), generator=torch.Generator().manual_seed(2))
    with torch.no_grad():
        _, idx0, _ = mod(x.to(DEV))
        new = torch.randn((1, 256, 64), generator=torch.Generator().manual_seed(77))
        # (a) in-place copy_: bumps the version counter
        mod._codebook.embeddings.copy_(new.to(DEV))
        q1, idx1, _ = mod(x.to(DEV))
    ref = oracle.vq_forward(x.reshape(1, -1, 64).numpy(), new.numpy())
    np.testing.assert_array_equal(idx1.cpu().numpy().reshape(-1), ref["idx"][0])
    np.testing.assert_array_equal(q1.cpu().numpy().reshape(-1, 64), ref["out"][0])
    assert not np.array_equal(idx0.cpu().numpy(), idx1.cpu().numpy())
    # (b) load_state_dict
    newer = torch.randn((1, 256, 64), generator=torch.Generator().manual_seed(78))
    sd = {k: v.clone() for k, v in mod.state_dict().items()}
    sd["_codebook.embeddings"] = newer.to(DEV)
    mod.load_state_dict(sd)
    with torch.no_grad():
        _, idx2, _ = mod(x.to(DEV))
    ref2 = oracle.vq_forward(x.reshape(1, -1, 64).numpy(), newer.numpy())
    np.testing.assert_array_equal(idx2.cpu().numpy().reshape(-1), ref2["idx"][0])
    # (c) a write the version counter cannot see + the documented invalidation
    newest = torch.randn((1, 256, 64), generator=torch.Generator().manual_seed(79))
    mod._codebook.embeddings.data.copy_(newest.to(DEV))
    mod._codebook.invalidate_packed()
    with torch.no_grad():
        _, idx3, _ = mod(x.to(DEV))
    ref3 = oracle.vq_forward(x.reshape(1, -1, 64).numpy(), newest.numpy())
    np.testing.assert_array_equal(idx3.cpu().numpy().reshape(-1), ref3["idx"][0])


def test_ema_step_invalidates_the_cache(oracle):
    """Two training forwards: the second must search the codebook the first one's EMA step wrote."""
    mod = _vq(decay=0.5, threshold_ema_dead_code=0).train()
    x = torch.randn((2, 300, 64), generator=torch.Generator().manual_seed(3))
    with torch.no_grad():
        mod(x.to(DEV))
        after_first = mod._codebook.embeddings.detach().cpu().clone()
        _, idx, _ = mod(x.to(DEV), freeze_codebook=True)
    ref = oracle.vq_forward(x.reshape(1, -1, 64).numpy(), after_first.numpy())
    np.testing.assert_array_equal(idx.cpu().numpy().reshape(-1), ref["idx"][0])


def test_residual_stack_cache_follows_layer_updates(oracle):
    mod = _rvq().eval()
    x = torch.randn((2, 150, 64), generator=torch.Generator().manual_seed(4))
    with torch.no_grad():
        mod(x.to(DEV))
        new2 = torch.randn((1, 128, 64), generator=torch.Generator().manual_seed(90)) * 0.5
        mod.layers[2]._codebook.embeddings.copy_(new2.to(DEV))
        q, idx, _ = mod(x.to(DEV))
    cbs = torch.stack([layer._codebook.embeddings[0] for layer in mod.layers]).cpu().numpy()
    ref = oracle.rvq_forward(x.reshape(-1, 64).numpy(), cbs)
    np.testing.assert_array_equal(idx.cpu().numpy().reshape(-1, 4), ref["idx"])
    np.testing.assert_array_equal(q.cpu().numpy().reshape(-1, 64), ref["out"])


def test_module_moved_between_devices_repacks():
    mod = _vq().eval()
    x = torch.randn((1, 64, 64), generator=torch.Generator().manual_seed(8)).to(DEV)
    with torch.no_grad():
        _, idx0, _ = mod(x)
        mod = mod.cpu().to(DEV)  # new storage, same values
        _, idx1, _ = mod(x)
    assert torch.equal(idx0, idx1)


@pytest.mark.parametrize("training", [False, True])
def test_stack_longer_than_one_launch_runs_layer_by_layer(oracle, training):
    """32 stages of dim 256 with the commitment loss exceed the fused launch's LDS budget (vq_max_fused_stages = 30):
    the module falls back to one launch per layer and still reproduces the oracle."""
    from vector_quantization import native

    Q, K, D = 32, 64, 256
    assert native.max_fused_stages(D, True) < Q <= native.max_fused_stages(D, False)
    mod = _rvq(dim=D, K=K, Q=Q, ema_update=False)
    mod.train(training)
    x = torch.randn((2, 40, D), generator=torch.Generator().manual_seed(9))
    with torch.no_grad():
        q, idx, losses = mod(x.to(DEV), freeze_codebook=True)
    cbs = torch.stack([layer._codebook.embeddings[0] for layer in mod.layers]).cpu().numpy()
    ref = oracle.rvq_forward(x.reshape(-1, D).numpy(), cbs, oracle.EUCLID, training=training)
    np.testing.assert_array_equal(idx.cpu().numpy().reshape(-1, Q), ref["idx"])
    np.testing.assert_allclose(q.cpu().numpy().reshape(-1, D), ref["out"], rtol=0, atol=1e-5)
    if training:
        np.testing.assert_allclose(losses.cpu().numpy().reshape(-1), ref["sq_err"] / x.numel(), rtol=1e-5, atol=1e-7)
