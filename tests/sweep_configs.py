"""Configuration lists shared by the container-only reference sweep (tests/golden/sweep_against_reference.py) and the GPU
sweep (tests/test_gpu_sweep.py): (class name, constructor kwargs with the CodebookParams kwargs under "cb", input shape,
forward kwargs).  ``mask=True`` / ``given_indices=True`` are placeholders the runners turn into tensors."""
from __future__ import annotations

import itertools

import torch


def forward_cases():
    cases = []
    noexp = dict(threshold_ema_dead_code=0)
    for heads, sep, cdim in [(1, False, None), (2, True, 16), (2, False, 16), (2, False, None)]:
        for cos in (False, True):
            cb = dict(dim=cdim or 32, codebook_size=40, use_cosine_sim=cos, **noexp)
            if cos:
                cb.update(transform_input="l2norm", weights_regularization="l2norm")
            for shape, cl in [((2, 30, 32), True), ((2, 32, 5, 6), False), ((7, 32), True)]:
                cases.append(("VectorQuantize", dict(dim=32, heads=heads, separate_codebook_per_head=sep, codebook_dim=cdim,
                                                     channel_last=cl, cb=cb), shape, {}))
    for shared, cdim, drop in itertools.product((False, True), (None, 16), (False, True)):
        ctor = dict(dim=32, num_quantizers=4, shared_codebook=shared, codebook_dim=cdim,
                    cb=dict(dim=cdim or 32, codebook_size=40, **noexp))
        fwd = {}
        if drop:
            ctor.update(quantize_dropout=True, quantize_dropout_cutoff_index=1)
            fwd = dict(rand_quantize_dropout_fixed_seed=3)
        cases.append(("ResidualVQ", ctor, (2, 30, 32), fwd))
        cases.append(("ResidualVQ", dict(ctor, cb=dict(ctor["cb"])), (2, 5, 6, 32), dict(fwd, return_all_codes=True)))
    # masks, similarity-consuming losses, cross entropy to given indices
    for heads, sep, cdim in [(1, False, None), (2, True, 16), (2, False, 16)]:
        base = dict(dim=32, heads=heads, separate_codebook_per_head=sep, codebook_dim=cdim)
        cbk = dict(dim=cdim or 32, codebook_size=40, **noexp)
        cases.append(("VectorQuantize", dict(base, cb=cbk), (2, 30, 32), dict(mask=True)))
        cases.append(("VectorQuantize", dict(base, commitment_use_cross_entropy_loss=True, cb=cbk), (2, 30, 32), {}))
        cases.append(("VectorQuantize", dict(base, commitment_use_cross_entropy_loss=True, cb=cbk), (2, 30, 32), dict(mask=True)))
        cases.append(("VectorQuantize", dict(base, codebook_diversity_loss_weight=0.3, codebook_diversity_temperature=2.0,
                                             cb=cbk), (2, 30, 32), {}))
        cases.append(("VectorQuantize", dict(base, cb=cbk), (2, 30, 32), dict(given_indices=True)))
    # remaining constructor options
    cbk = dict(dim=16, codebook_size=40, **noexp)
    cases.append(("VectorQuantize", dict(dim=32, codebook_dim=16, layernorm_after_project_in=True, cb=cbk), (2, 30, 32), {}))
    cases.append(("VectorQuantize", dict(dim=32, codebook_dim=16, commitment_weight=0.0, cb=cbk), (2, 30, 32), {}))
    cases.append(("VectorQuantize", dict(dim=32, codebook_dim=16, commitment_weight=2.5, cb=cbk), (2, 30, 32),
                  dict(return_loss_breakdown=True)))
    cases.append(("VectorQuantize", dict(dim=32, heads=4, cb=dict(dim=32, codebook_size=40, **noexp)), (3, 32), {}))
    cases.append(("ResidualVQ", dict(dim=32, num_quantizers=3, commitment_weight=0.25, cb=dict(dim=32, codebook_size=40, **noexp)),
                  (2, 30, 32), {}))
    for groups in (2, 4):
        cases.append(("GroupedResidualVQ", dict(dim=32, groups=groups, num_quantizers=3,
                                                cb=dict(dim=32 // groups, codebook_size=24, **noexp)), (2, 30, 32), {}))
    # masks on the stacks, learnable codebook with a mask, EMA parameter variations
    cases.append(("ResidualVQ", dict(dim=32, num_quantizers=3, cb=dict(dim=32, codebook_size=40, **noexp)), (2, 30, 32), dict(mask=True)))
    cases.append(("ResidualVQ", dict(dim=32, num_quantizers=3, shared_codebook=True, cb=dict(dim=32, codebook_size=40, **noexp)),
                  (2, 30, 32), dict(mask=True)))
    cases.append(("GroupedResidualVQ", dict(dim=32, groups=2, num_quantizers=2, cb=dict(dim=16, codebook_size=24, **noexp)),
                  (2, 30, 32), dict(mask=True)))
    cases.append(("VectorQuantize", dict(dim=32, cb=dict(dim=32, codebook_size=40, learnable_codebook=True, ema_update=False, **noexp)),
                  (2, 30, 32), dict(mask=True)))
    cases.append(("VectorQuantize", dict(dim=32, cb=dict(dim=32, codebook_size=40, decay=0.5, eps_for_smoothing=1e-3, **noexp)),
                  (2, 30, 32), {}))
    cases.append(("VectorQuantize", dict(dim=32, cb=dict(dim=32, codebook_size=40, ema_update=False, **noexp)), (2, 30, 32), {}))
    # cosine similarity through the stacks (with and without the l2norm transforms), cosine + similarity-consuming losses
    for l2 in (False, True):
        cb = dict(dim=32, codebook_size=40, use_cosine_sim=True, **noexp)
        if l2:
            cb.update(transform_input="l2norm", weights_regularization="l2norm")
        cases.append(("ResidualVQ", dict(dim=32, num_quantizers=3, cb=dict(cb)), (2, 30, 32), {}))
        cases.append(("ResidualVQ", dict(dim=32, num_quantizers=3, shared_codebook=True, cb=dict(cb)), (2, 30, 32), {}))
        cases.append(("GroupedResidualVQ", dict(dim=32, groups=2, num_quantizers=2, cb=dict(cb, dim=16)), (2, 30, 32), {}))
    cases.append(("VectorQuantize", dict(dim=32, heads=2, codebook_dim=16, separate_codebook_per_head=True,
                                         commitment_use_cross_entropy_loss=True,
                                         cb=dict(dim=16, codebook_size=40, use_cosine_sim=True, **noexp)), (2, 30, 32), {}))
    cases.append(("VectorQuantize", dict(dim=32, codebook_diversity_loss_weight=0.2, codebook_diversity_temperature=3.0,
                                         cb=dict(dim=32, codebook_size=40, use_cosine_sim=True, transform_input="l2norm",
                                                 weights_regularization="l2norm", **noexp)), (2, 30, 32), {}))
    # rows wider than 512 dims (no width limit in the reference; sliced sweep on the device) -- incl. the losses
    # whose fused kernels stop at 512 dims and fall back to row chunks of the similarity matrix there
    wide = dict(dim=640, codebook_size=40, **noexp)
    cases.append(("VectorQuantize", dict(dim=640, cb=dict(wide)), (2, 30, 640), {}))
    cases.append(("VectorQuantize", dict(dim=640, cb=dict(wide)), (2, 30, 640), dict(mask=True)))
    cases.append(("VectorQuantize", dict(dim=640, commitment_use_cross_entropy_loss=True, cb=dict(wide)), (2, 30, 640), {}))
    cases.append(("VectorQuantize", dict(dim=640, codebook_diversity_loss_weight=0.3, codebook_diversity_temperature=2.0,
                                         cb=dict(wide)), (2, 30, 640), {}))
    cases.append(("VectorQuantize", dict(dim=600, cb=dict(dim=600, codebook_size=40, use_cosine_sim=True, **noexp)), (2, 30, 600), {}))
    cases.append(("ResidualVQ", dict(dim=576, num_quantizers=3, cb=dict(dim=576, codebook_size=40, **noexp)), (2, 30, 576), {}))
    cases.append(("VectorQuantize", dict(dim=640, cb=dict(wide)), (2, 30, 640), dict(given_indices=True)))
    cases.append(("GroupedResidualVQ", dict(dim=1280, groups=2, num_quantizers=2, cb=dict(dim=640, codebook_size=24, **noexp)),
                  (2, 30, 1280), {}))
    return cases


def autograd_cases():
    learn = dict(learnable_codebook=True, ema_update=False)
    grad_cases = [c for c in forward_cases() if "given_indices" not in c[3] and not c[1].get("quantize_dropout")][::3]
    for cdim in (None, 16):
        grad_cases.append(("VectorQuantize", dict(dim=32, codebook_dim=cdim, cb=dict(dim=cdim or 32, codebook_size=40, **learn)),
                           (2, 30, 32), {}))
        grad_cases.append(("VectorQuantize", dict(dim=32, codebook_dim=cdim, sync_update_v=0.3,
                                                  cb=dict(dim=cdim or 32, codebook_size=40, **learn)), (2, 30, 32), {}))
        grad_cases.append(("ResidualVQ", dict(dim=32, num_quantizers=3, codebook_dim=cdim,
                                              cb=dict(dim=cdim or 32, codebook_size=40, **learn)), (2, 30, 32), {}))
    return grad_cases


def materialise_forward_kwargs(ctor, cb_kw, x, fwd):
    """Turn the placeholders into tensors (same seeds on every side of a comparison)."""
    kw = dict(fwd)
    if "mask" in kw:
        lengths = torch.tensor([max(1, x.shape[1] // (1 + i % 2)) for i in range(x.shape[0])])
        kw["mask"] = torch.arange(x.shape[1])[None, :] < lengths[:, None]
    if kw.pop("given_indices", False):
        heads = ctor.get("heads", 1)
        n = (x.numel() // (x.shape[0] * x.shape[-1]) if ctor.get("channel_last", True)
             else x.numel() // (x.shape[0] * x.shape[1]))
        kw["indices"] = torch.randint(0, cb_kw["codebook_size"], (x.shape[0], n, heads) if heads > 1 else (x.shape[0], n),
                                      generator=torch.Generator().manual_seed(3))
    return kw
