"""Generate golden fixtures by IMPORTING THE REFERENCE (this container only; never runs on the GPU box).

    python tests/golden/make_golden.py            # writes tests/golden/data/<case>.npz

The reference (/root/reference, read-only) is pure Python on top of PyTorch.  Its one missing
dependency, ``einx`` (pinned 0.3.0 in uv.lock, not installed, no network), is used for exactly one
call -- ``get_at("q [c] d, b n q -> q b n d", codebooks, indices)`` at residual_vq.py:117 -- which is
provided here by a 4-line local stand-in registered in sys.modules before the import.

Fixtures are data only: seeds/config, the expected indices, losses, checksums and sampled rows of
the quantized output (plus any randomly initialised projection weights).  No reference source is
copied.  Metadata records torch version and thread count (goldens come from torch's CPU/MKL path).
"""
from __future__ import annotations

import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

from cases import CASES, LOSS_CASES  # noqa: E402
from gen import CB_SEED, checksum, l2norm, make_codebook, make_rvq_codebooks, make_x, poison_, seeded_projection_  # noqa: E402


def _import_reference():
    einx = types.ModuleType("einx")

    def get_at(pattern, codebooks, indices):
        assert pattern == "q [c] d, b n q -> q b n d"
        return torch.stack([codebooks[q][indices[..., q]] for q in range(codebooks.shape[0])])

    einx.get_at = get_at
    sys.modules["einx"] = einx
    sys.path.insert(0, "/root/reference")
    import vector_quantization as ref  # noqa
    from vector_quantization import codebooks as ref_cb  # noqa

    return ref, ref_cb


def make_mask(b, n):
    mask = torch.zeros(b, n, dtype=torch.bool)
    for i in range(b):
        mask[i, : max(1, n // (i + 1))] = True
    return mask


def sample_rows(t: torch.Tensor, last_dim: int):
    flat = t.detach().reshape(-1, last_dim)
    n = flat.shape[0]
    rows = sorted(set(list(range(min(8, n))) + list(range(max(0, n - 8), n))))
    return np.asarray(rows, dtype=np.int64), flat[rows].numpy().copy()


def run_vq(ref, ref_cb, c):
    dim, K = c["dim"], c["K"]
    heads = c.get("heads", 1)
    separate = c.get("separate_codebook_per_head", False)
    codebook_dim = c.get("codebook_dim", None)
    d = codebook_dim if codebook_dim is not None else dim
    h = heads if separate else 1
    use_cos = c.get("use_cosine_sim", False)
    cb_extra = dict(c.get("cb_extra", {}))
    if "kmeans_iter" in cb_extra:
        cb_extra["kmeans_params"] = ref_cb.KmeansParameters(iter=cb_extra.pop("kmeans_iter"), sync=False)
    params = ref_cb.CodebookParams(
        dim=d, codebook_size=K, use_cosine_sim=use_cos,
        transform_input=c.get("transform_input", "identity"),
        weights_regularization=c.get("weights_regularization", "identity"),
        **cb_extra,
    )
    torch.manual_seed(777)
    mod = ref.VectorQuantize(dim=dim, codebook_params=params, codebook_dim=codebook_dim, heads=heads,
                             separate_codebook_per_head=separate, channel_last=c.get("channel_last", True))
    cb = make_codebook(h, K, d, c["cls"])
    if c.get("weights_regularization", "identity") == "l2norm":
        cb = l2norm(cb)
    x = make_x(c["x_shape"], c["cls"])
    poison_(x, cb, c.get("nonfinite"))
    with torch.no_grad():
        mod._codebook.embeddings.copy_(cb)
        mod._codebook.embed_avg.copy_(cb)
    if c.get("seeded_proj", False):
        seeded_projection_(mod)
    stash = {}

    def hook(_m, _inp, out):
        sim = out[2]
        stash["best"] = sim.max(dim=-1).values.detach().clone()

    mod._codebook.register_forward_hook(hook)
    kwargs = {}
    mask = None
    if c.get("mask", False):
        mask = make_mask(x.shape[0], x.shape[1])
        kwargs["mask"] = mask
    if c["training"]:
        mod.train()
        kwargs["freeze_codebook"] = c.get("freeze_codebook", True)
    else:
        mod.eval()
    if "forward_seed" in c:
        torch.manual_seed(c["forward_seed"])
    with torch.no_grad():
        q, idx, loss = mod(x, **kwargs)
    arrays = dict(idx=idx.numpy().astype(np.int32), loss=loss.detach().numpy().astype(np.float32),
                  ref_best=stash["best"].numpy().astype(np.float32))
    if c["training"] and not c.get("freeze_codebook", True):
        arrays["ema_embeddings"] = mod._codebook.embeddings.detach().numpy().copy()
        arrays["ema_embed_avg"] = mod._codebook.embed_avg.detach().numpy().copy()
        arrays["ema_cluster_size"] = mod._codebook.cluster_size.detach().numpy().copy()
    feat_last = q.shape[-1] if c.get("channel_last", True) else None
    qcl = q if c.get("channel_last", True) else q.movedim(1, -1)
    rows, vals = sample_rows(qcl, qcl.shape[-1])
    arrays["q_rows"], arrays["q_vals"] = rows, vals
    if q.numel() <= 1 << 16:
        arrays["q_full"] = q.detach().numpy().copy()
    if mod.has_projections and not c.get("seeded_proj", False):
        lin_in = mod.project_in if isinstance(mod.project_in, torch.nn.Linear) else mod.project_in[0]
        arrays["proj_in_w"] = lin_in.weight.detach().numpy().copy()
        arrays["proj_in_b"] = lin_in.bias.detach().numpy().copy()
        arrays["proj_out_w"] = mod.project_out.weight.detach().numpy().copy()
        arrays["proj_out_b"] = mod.project_out.bias.detach().numpy().copy()
    meta = dict(x_checksum=checksum(x), cb_checksum=checksum(cb), q_checksum=checksum(q),
                q_shape=list(q.shape), idx_shape=list(idx.shape))
    del feat_last
    return arrays, meta


def q_weights(shape):
    """Seeded weights for the cases that also backpropagate through the quantized output."""
    return torch.randn(tuple(shape), generator=torch.Generator().manual_seed(4242))


def given_indices(c, rows_shape, K):
    """Teacher-forcing targets for the ``indices=`` cases (seeded; shared with tests/build_case.py)."""
    g = torch.Generator().manual_seed(99)
    t = torch.randint(0, K, rows_shape, generator=g)
    if c.get("ignore_some", False):
        t.view(-1)[::5] = -1
    return t


def run_vqloss(ref, ref_cb, c):
    """Similarity-consuming losses: loss value(s), returned tensors and d loss / d x of the reference."""
    dim, K = c["dim"], c["K"]
    heads = c.get("heads", 1)
    separate = c.get("separate_codebook_per_head", False)
    codebook_dim = c.get("codebook_dim", None)
    d = codebook_dim if codebook_dim is not None else dim
    h = heads if separate else 1
    channel_last = c.get("channel_last", True)
    params = ref_cb.CodebookParams(
        dim=d, codebook_size=K, use_cosine_sim=c.get("use_cosine_sim", False),
        transform_input=c.get("transform_input", "identity"),
        weights_regularization=c.get("weights_regularization", "identity"), **c.get("cb_extra", {}))
    torch.manual_seed(777)
    extra = dict(c.get("vq_extra", {}))
    if "inplace_sgd_lr" in c:
        extra["in_place_codebook_optimizer"] = lambda params: torch.optim.SGD(params, lr=c["inplace_sgd_lr"])
    mod = ref.VectorQuantize(dim=dim, codebook_params=params, codebook_dim=codebook_dim, heads=heads,
                             separate_codebook_per_head=separate, channel_last=channel_last, **extra)
    assert not mod.has_projections
    cb = make_codebook(h, K, d, c["cls"])
    if c.get("weights_regularization", "identity") == "l2norm":
        cb = l2norm(cb)
    with torch.no_grad():
        mod._codebook.embeddings.copy_(cb)
        mod._codebook.embed_avg.copy_(cb)
    x = make_x(c["x_shape"], c["cls"]).requires_grad_(True)
    kwargs = {}
    if c.get("mask", False):
        kwargs["mask"] = make_mask(x.shape[0], x.shape[1])
    if c["training"]:
        mod.train()
        kwargs["freeze_codebook"] = c.get("freeze_codebook", True)
    else:
        mod.eval()
    arrays = {}
    if c.get("given_indices", False):
        b = x.shape[0]
        n = x.numel() // (b * dim)
        shape = (b, n, heads) if heads > 1 else (b, n)
        kwargs["indices"] = given_indices(c, shape, K)
        q, loss = mod(x, **kwargs)
        idx = None
    else:
        q, idx, loss, breakdown = mod(x, return_loss_breakdown=True, **kwargs)
        arrays["idx"] = idx.numpy().astype(np.int32)
        arrays["breakdown"] = np.asarray([float(v) for v in breakdown], dtype=np.float32)
    objective = loss.sum()
    if c.get("backprop_q", False):
        objective = objective + (q * q_weights(q.shape)).sum()
    objective.backward()
    grad = x.grad.detach()
    arrays["loss"] = loss.detach().numpy().astype(np.float32)
    arrays["q_shape"] = np.asarray(q.shape, dtype=np.int64)
    flat_q = q.detach().reshape(-1, q.shape[-1])
    rows, vals = sample_rows(flat_q, flat_q.shape[-1])
    arrays["q_rows"], arrays["q_vals"] = rows, vals
    gcl = grad if channel_last else grad.movedim(1, -1)
    rows, vals = sample_rows(gcl, gcl.shape[-1])
    arrays["g_rows"], arrays["g_vals"] = rows, vals
    if grad.numel() <= 1 << 16:
        arrays["g_full"] = grad.numpy().copy()
    if c["training"] and not c.get("freeze_codebook", True):
        arrays["ema_embeddings"] = mod._codebook.embeddings.detach().numpy().copy()
        if mod._codebook.embeddings.grad is not None:
            arrays["cb_grad"] = mod._codebook.embeddings.grad.detach().numpy().copy()
    meta = dict(x_checksum=checksum(x.detach()), cb_checksum=checksum(cb), q_checksum=checksum(q.detach()),
                g_checksum=checksum(grad), q_shape=list(q.shape))
    return arrays, meta


def run_orthogonal(ref):
    """utils/losses.py:23-28 on seeded codebooks (the module path crashes in the fork: ``_codebook.embed``)."""
    from vector_quantization.utils.losses import orthogonal_loss_fn

    arrays = {}
    for i, (h, k, d) in enumerate([(1, 256, 64), (4, 100, 32), (1, 1024, 256)]):
        cb = make_codebook(h, k, d, "S", seed=CB_SEED + i).requires_grad_(True)
        v = orthogonal_loss_fn(cb)
        v.backward()
        arrays[f"shape{i}"] = np.asarray([h, k, d], dtype=np.int64)
        arrays[f"value{i}"] = np.asarray(float(v), dtype=np.float64)
        arrays[f"grad_rows{i}"] = cb.grad[:, :4].numpy().copy()
    return arrays, dict(n=3)


def run_rvq(ref, ref_cb, c):
    dim, K, Q = c["dim"], c["K"], c["Q"]
    params = ref_cb.CodebookParams(dim=dim, codebook_size=K, **c.get("cb_extra", {}))
    shared = c.get("shared_codebook", False)
    mod = ref.ResidualVQ(dim=dim, num_quantizers=Q, codebook_params=params, shared_codebook=shared,
                         **c.get("vq_extra", {}), **c.get("rvq_extra", {}))
    cbs = make_rvq_codebooks(Q, K, dim, c["cls"])
    x = make_x(c["x_shape"], c["cls"])
    poison_(x, cbs, c.get("nonfinite"))
    with torch.no_grad():
        for i, layer in enumerate(mod.layers):
            layer._codebook.embeddings.copy_(cbs[0 if shared else i][None])
            layer._codebook.embed_avg.copy_(cbs[0 if shared else i][None])
    kwargs = {}
    if c["training"]:
        mod.train()
        kwargs["freeze_codebook"] = c.get("freeze_codebook", True)
    else:
        mod.eval()
    if c.get("return_all_codes", False):
        kwargs["return_all_codes"] = True
    kwargs.update(c.get("fwd_extra", {}))
    with torch.no_grad():
        out = mod(x, **kwargs)
    q, idx, losses = out[:3]
    arrays = dict(idx=idx.numpy().astype(np.int32), loss=losses.detach().numpy().astype(np.float32))
    rows, vals = sample_rows(q, q.shape[-1])
    arrays["q_rows"], arrays["q_vals"] = rows, vals
    if q.numel() <= 1 << 16:
        arrays["q_full"] = q.detach().numpy().copy()
    if len(out) > 3:
        arrays["all_codes"] = out[3].detach().numpy().copy()
    if c["training"] and not c.get("freeze_codebook", True):
        arrays["ema_embeddings"] = torch.stack([l._codebook.embeddings for l in mod.layers]).detach().numpy().copy()
        arrays["ema_embed_avg"] = torch.stack([l._codebook.embed_avg for l in mod.layers]).detach().numpy().copy()
        arrays["ema_cluster_size"] = torch.stack([l._codebook.cluster_size for l in mod.layers]).detach().numpy().copy()
    meta = dict(x_checksum=checksum(x), cb_checksum=checksum(cbs), q_checksum=checksum(q),
                q_shape=list(q.shape), idx_shape=list(idx.shape))
    return arrays, meta


def run_grvq(ref, ref_cb, c):
    dim, K, Q, G = c["dim"], c["K"], c["Q"], c["groups"]
    d = dim // G
    params = ref_cb.CodebookParams(dim=d, codebook_size=K, **c.get("cb_extra", {}))
    mod = ref.GroupedResidualVQ(dim=dim, groups=G, num_quantizers=Q, codebook_params=params)
    all_cbs = []
    with torch.no_grad():
        for g, rvq in enumerate(mod.rvqs):
            cbs = make_rvq_codebooks(Q, K, d, c["cls"], seed=CB_SEED + 100 * g)
            all_cbs.append(cbs)
            for i, layer in enumerate(rvq.layers):
                layer._codebook.embeddings.copy_(cbs[i][None])
                layer._codebook.embed_avg.copy_(cbs[i][None])
    x = make_x(c["x_shape"], c["cls"])
    kwargs = {}
    if c["training"]:
        mod.train()
        kwargs["freeze_codebook"] = c.get("freeze_codebook", True)
    else:
        mod.eval()
    with torch.no_grad():
        q, idx, losses = mod(x, **kwargs)
    arrays = dict(idx=idx.numpy().astype(np.int32), loss=losses.detach().numpy().astype(np.float32),
                  q_full=q.detach().numpy().copy())
    rows, vals = sample_rows(q, q.shape[-1])
    arrays["q_rows"], arrays["q_vals"] = rows, vals
    if c["training"] and not c.get("freeze_codebook", True):
        layers = [l for rvq in mod.rvqs for l in rvq.layers]
        arrays["ema_embeddings"] = torch.stack([l._codebook.embeddings for l in layers]).detach().numpy().copy()
        arrays["ema_embed_avg"] = torch.stack([l._codebook.embed_avg for l in layers]).detach().numpy().copy()
        arrays["ema_cluster_size"] = torch.stack([l._codebook.cluster_size for l in layers]).detach().numpy().copy()
    meta = dict(x_checksum=checksum(x), cb_checksum=checksum(torch.stack(all_cbs)), q_checksum=checksum(q),
                q_shape=list(q.shape), idx_shape=list(idx.shape))
    return arrays, meta


def main():
    ref, ref_cb = _import_reference()
    out_dir = os.path.join(HERE, "data")
    os.makedirs(out_dir, exist_ok=True)
    only = set(sys.argv[1:])
    if not only or "orthogonal_fn" in only:
        arrays, meta = run_orthogonal(ref)
        meta.update(torch_version=torch.__version__, generator="tests/golden/make_golden.py importing /root/reference @ 2024_10_08")
        arrays["meta_json"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
        np.savez_compressed(os.path.join(out_dir, "orthogonal_fn.npz"), **arrays)
        print("orthogonal_fn", [float(arrays[f"value{i}"]) for i in range(3)])
    for c in CASES + LOSS_CASES:
        if only and c["name"] not in only:
            continue
        fn = dict(vq=run_vq, rvq=run_rvq, grvq=run_grvq, vqloss=run_vqloss)[c["kind"]]
        arrays, meta = fn(ref, ref_cb, c)
        meta.update(case=c, torch_version=torch.__version__, num_threads=torch.get_num_threads(),
                    blas="mkl" if torch.backends.mkl.is_available() else "other",
                    generator="tests/golden/make_golden.py importing /root/reference @ 2024_10_08")
        arrays["meta_json"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
        path = os.path.join(out_dir, c["name"] + ".npz")
        np.savez_compressed(path, **arrays)
        print(f"{c['name']:>20s}  idx{tuple(arrays['idx'].shape) if 'idx' in arrays else ()}  loss={arrays['loss'].ravel()[:3]}  "
              f"{os.path.getsize(path) / 1024:.1f} KiB")


if __name__ == "__main__":
    main()
