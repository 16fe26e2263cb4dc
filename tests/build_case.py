"""Build the drop-in module + inputs for a golden case (the 'harness counterpart' of SURVEY 8c)."""
from __future__ import annotations

import torch

from gen import CB_SEED, l2norm, make_codebook, make_rvq_codebooks, make_x, poison_, seeded_projection_


def make_mask(b, n):
    mask = torch.zeros(b, n, dtype=torch.bool)
    for i in range(b):
        mask[i, : max(1, n // (i + 1))] = True
    return mask


def q_weights(shape):
    """Same seeded weights as tests/golden/make_golden.py:q_weights."""
    return torch.randn(tuple(shape), generator=torch.Generator().manual_seed(4242))


def given_indices(case, shape, K):
    """Same seeded teacher-forcing targets as tests/golden/make_golden.py:given_indices."""
    g = torch.Generator().manual_seed(99)
    t = torch.randint(0, K, shape, generator=g)
    if case.get("ignore_some", False):
        t.view(-1)[::5] = -1
    return t


def build(case, arrays=None, device="cpu"):
    """-> (module, x, forward_kwargs, codebook_tensor)."""
    import vector_quantization as vq
    from vector_quantization.codebooks import CodebookParams

    kind = case["kind"]
    x = make_x(case["x_shape"], case["cls"])
    kwargs = {}
    if kind in ("vq", "vqloss"):
        dim, K = case["dim"], case["K"]
        heads = case.get("heads", 1)
        separate = case.get("separate_codebook_per_head", False)
        codebook_dim = case.get("codebook_dim", None)
        d = codebook_dim if codebook_dim is not None else dim
        h = heads if separate else 1
        cb_extra = dict(case.get("cb_extra", {}))
        if "kmeans_iter" in cb_extra:
            from vector_quantization.codebooks import KmeansParameters

            cb_extra["kmeans_params"] = KmeansParameters(iter=cb_extra.pop("kmeans_iter"), sync=False)
        params = CodebookParams(dim=d, codebook_size=K, use_cosine_sim=case.get("use_cosine_sim", False),
                                transform_input=case.get("transform_input", "identity"),
                                weights_regularization=case.get("weights_regularization", "identity"),
                                **cb_extra)
        extra = dict(case.get("vq_extra", {}))
        if "inplace_sgd_lr" in case:
            extra["in_place_codebook_optimizer"] = lambda params: torch.optim.SGD(params, lr=case["inplace_sgd_lr"])
        mod = vq.VectorQuantize(dim=dim, codebook_params=params, codebook_dim=codebook_dim, heads=heads,
                                separate_codebook_per_head=separate, channel_last=case.get("channel_last", True),
                                **extra)
        cb = make_codebook(h, K, d, case["cls"])
        if case.get("weights_regularization", "identity") == "l2norm":
            cb = l2norm(cb)
        poison_(x, cb, case.get("nonfinite"))
        with torch.no_grad():
            mod._codebook.embeddings.copy_(cb)
            mod._codebook.embed_avg.copy_(cb)
            if mod.has_projections and case.get("seeded_proj", False):
                seeded_projection_(mod)
            elif mod.has_projections:
                assert arrays is not None
                mod.project_in.weight.copy_(torch.from_numpy(arrays["proj_in_w"]))
                mod.project_in.bias.copy_(torch.from_numpy(arrays["proj_in_b"]))
                mod.project_out.weight.copy_(torch.from_numpy(arrays["proj_out_w"]))
                mod.project_out.bias.copy_(torch.from_numpy(arrays["proj_out_b"]))
        if case.get("mask", False):
            kwargs["mask"] = make_mask(x.shape[0], x.shape[1]).to(device)
        if case.get("given_indices", False):
            b = x.shape[0]
            n = x.numel() // (b * dim)
            kwargs["indices"] = given_indices(case, (b, n, heads) if heads > 1 else (b, n), K).to(device)
    elif kind == "rvq":
        dim, K, Q = case["dim"], case["K"], case["Q"]
        shared = case.get("shared_codebook", False)
        mod = vq.ResidualVQ(dim=dim, num_quantizers=Q,
                            codebook_params=CodebookParams(dim=dim, codebook_size=K, **case.get("cb_extra", {})),
                            shared_codebook=shared, **case.get("vq_extra", {}), **case.get("rvq_extra", {}))
        cb = make_rvq_codebooks(Q, K, dim, case["cls"])
        poison_(x, cb, case.get("nonfinite"))
        with torch.no_grad():
            for i, layer in enumerate(mod.layers):
                layer._codebook.embeddings.copy_(cb[0 if shared else i][None])
                layer._codebook.embed_avg.copy_(cb[0 if shared else i][None])
        if case.get("return_all_codes", False):
            kwargs["return_all_codes"] = True
        kwargs.update(case.get("fwd_extra", {}))
    elif kind == "grvq":
        dim, K, Q, G = case["dim"], case["K"], case["Q"], case["groups"]
        d = dim // G
        mod = vq.GroupedResidualVQ(dim=dim, groups=G, num_quantizers=Q,
                                   codebook_params=CodebookParams(dim=d, codebook_size=K, **case.get("cb_extra", {})))
        cbs = []
        with torch.no_grad():
            for g, rvq in enumerate(mod.rvqs):
                c = make_rvq_codebooks(Q, K, d, case["cls"], seed=CB_SEED + 100 * g)
                cbs.append(c)
                for i, layer in enumerate(rvq.layers):
                    layer._codebook.embeddings.copy_(c[i][None])
                    layer._codebook.embed_avg.copy_(c[i][None])
        cb = torch.stack(cbs)
    else:
        raise ValueError(kind)
    if case["training"]:
        mod.train()
        kwargs["freeze_codebook"] = case.get("freeze_codebook", True)
    else:
        mod.eval()
    return mod.to(device), x.to(device), kwargs, cb
