"""Compare a module's outputs with a golden fixture (shared by the CPU host-logic and the GPU tests)."""
from __future__ import annotations

import numpy as np
import torch

Q_TOL = 1e-5  # north_star: quantized vectors and commitment loss within 1e-5 (fp32)


def certify_near_ties(x_rows, cb, got_idx, ref_idx, metric_dot=False, ulps=8):
    """Every index mismatch must be a near-tie: in fp64 the two candidates' scores differ by < `ulps` fp32 ulps."""
    bad = np.argwhere(got_idx != ref_idx)
    worst = 0.0
    for m in bad[:, 0] if bad.ndim == 2 else bad:
        xv = x_rows[m].astype(np.float64)
        a, b = cb[got_idx[m]].astype(np.float64), cb[ref_idx[m]].astype(np.float64)
        if metric_dot:
            sa, sb = float(xv @ a), float(xv @ b)
        else:
            sa, sb = float(np.sqrt(((xv - a) ** 2).sum())), float(np.sqrt(((xv - b) ** 2).sum()))
        scale = max(abs(sa), abs(sb), 1e-30)
        # the fp32 GEMM form cancels |x|^2 + |c|^2 - 2xc: its rounding error is relative to |x|^2 + |c|^2
        mag = (xv ** 2).sum() + max((a ** 2).sum(), (b ** 2).sum())
        tol = ulps * np.spacing(np.float32(mag)) / (2 * max(scale, 1e-30)) if not metric_dot else \
            ulps * np.spacing(np.float32(np.abs(xv * a).sum()))
        worst = max(worst, abs(sa - sb) / max(tol, 1e-300))
        assert abs(sa - sb) <= tol, f"row {m}: idx {got_idx[m]} vs {ref_idx[m]} is not a near-tie: {sa} vs {sb} (tol {tol})"
    return len(bad), worst


def compare(case, arrays, meta, outputs, x, cb, mod=None):
    quantize, idx, loss = outputs[:3]
    quantize, idx, loss = quantize.detach().cpu(), idx.detach().cpu(), loss.detach().cpu()
    assert list(quantize.shape) == meta["q_shape"], (quantize.shape, meta["q_shape"])
    assert list(idx.shape) == meta["idx_shape"], (idx.shape, meta["idx_shape"])
    assert idx.dtype == torch.int64
    ref_idx = arrays["idx"].astype(np.int64)
    got_idx = idx.numpy()
    n_mismatch = int((got_idx != ref_idx).sum())
    if case["cls"] == "R" or case["name"] == "tiny_direct":
        # near-tie heavy class (or ATen's small-size direct kernel): report, and require the rest to agree
        frac = n_mismatch / got_idx.size
        assert frac < 0.02, f"{n_mismatch} of {got_idx.size} indices differ"
        if n_mismatch and case["kind"] == "vq":
            # every mismatch must be a certified near-tie, head by head, on the rows the search actually saw (channel-last,
            # after the module's own input projection and row transform)
            heads = case.get("heads", 1)
            xs = x.detach().cpu()
            if not case.get("channel_last", True):
                xs = xs.movedim(1, -1)
            xs = xs.reshape(xs.shape[0], -1, xs.shape[-1])
            if mod is not None:
                with torch.no_grad():
                    dev = mod._codebook.embeddings.device
                    xs = mod.project_in(xs.to(dev)).cpu() if mod.has_projections else xs
                    xs = mod._codebook.transform_input(xs.reshape(*xs.shape[:-1], heads, -1)).reshape(xs.shape)
            d = xs.shape[-1] // heads
            rows = xs.reshape(-1, heads, d).numpy()
            gi, ri = got_idx.reshape(-1, heads), ref_idx.reshape(-1, heads)
            cbn = cb.cpu().numpy()
            for hd in range(heads):
                certify_near_ties(rows[:, hd], cbn[hd if cbn.shape[0] > 1 else 0], gi[:, hd], ri[:, hd],
                                  case.get("use_cosine_sim", False))
    else:
        assert n_mismatch == 0, f"{n_mismatch} of {got_idx.size} indices differ from the reference"
    assert loss.shape == torch.Size(arrays["loss"].shape), (loss.shape, arrays["loss"].shape)
    np.testing.assert_allclose(loss.numpy(), arrays["loss"], atol=Q_TOL, rtol=1e-5)
    if n_mismatch == 0:
        channel_last = case.get("channel_last", True)
        qcl = quantize if channel_last else quantize.movedim(1, -1)
        flat = qcl.reshape(-1, qcl.shape[-1]).numpy()
        np.testing.assert_allclose(flat[arrays["q_rows"]], arrays["q_vals"], atol=Q_TOL, rtol=0)
        if "q_full" in arrays:
            np.testing.assert_allclose(quantize.numpy(), arrays["q_full"], atol=Q_TOL, rtol=0)
        s = float(quantize.double().sum())
        if np.isfinite(meta["q_checksum"][0]):  # (non-finite cases: the element-wise comparisons above are the check)
            assert abs(s - meta["q_checksum"][0]) <= 1e-5 * max(1.0, meta["q_checksum"][1]), (s, meta["q_checksum"])
        else:
            assert not np.isfinite(s) or "nonfinite" in case
    if "ema_embeddings" in arrays and mod is not None:
        if case["kind"] == "grvq":
            layers = [l for rvq in mod.rvqs for l in rvq.layers]
            emb = torch.stack([l._codebook.embeddings for l in layers])
            avg = torch.stack([l._codebook.embed_avg for l in layers])
            cs = torch.stack([l._codebook.cluster_size for l in layers])
        elif case["kind"] == "rvq":
            emb = torch.stack([l._codebook.embeddings for l in mod.layers])
            avg = torch.stack([l._codebook.embed_avg for l in mod.layers])
            cs = torch.stack([l._codebook.cluster_size for l in mod.layers])
        else:
            emb, avg, cs = mod._codebook.embeddings, mod._codebook.embed_avg, mod._codebook.cluster_size
        # training-state step: statistics are sums over hundreds of rows -> relative tolerance on fp32 order
        np.testing.assert_allclose(cs.detach().cpu().numpy(), arrays["ema_cluster_size"], rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(avg.detach().cpu().numpy(), arrays["ema_embed_avg"], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(emb.detach().cpu().numpy(), arrays["ema_embeddings"], rtol=1e-4, atol=1e-5)
    if "all_codes" in arrays and len(outputs) > 3:
        np.testing.assert_allclose(outputs[3].detach().cpu().numpy(), arrays["all_codes"], atol=Q_TOL, rtol=0)
    return n_mismatch


# loss values: 1e-5 (north_star tolerance).  Gradients: 1e-5 relative to the largest gradient entry.  The diversity loss
# multiplies fp32 distances by its temperature before the softmax, so rounding differences between the reference's
# MKL sgemm and the k-ordered chain are amplified by that factor: tolerance scales with max(1, temperature).
def compare_loss(case, arrays, meta, mod, x, kwargs):
    """Run ``mod`` on ``x`` (requires_grad), backprop the loss, compare with the golden of a similarity-consuming case."""
    x = x.detach().clone().requires_grad_(True)
    temp = case.get("vq_extra", {}).get("codebook_diversity_temperature", 100.0) \
        if case.get("vq_extra", {}).get("codebook_diversity_loss_weight", 0.0) > 0 else 1.0
    amp = max(1.0, temp)
    if case.get("given_indices", False):
        q, loss = mod(x, **kwargs)
        assert loss.ndim == 0
    else:
        q, idx, loss, breakdown = mod(x, return_loss_breakdown=True, **kwargs)
        np.testing.assert_array_equal(idx.detach().cpu().numpy(), arrays["idx"].astype(np.int64))
        got_b = np.asarray([float(v.detach()) for v in breakdown], dtype=np.float32)
        np.testing.assert_allclose(got_b, arrays["breakdown"], rtol=1e-5 * amp, atol=1e-5 * amp)
    objective = loss.sum()
    if case.get("backprop_q", False):
        from build_case import q_weights

        objective = objective + (q * q_weights(q.shape).to(q.device)).sum()
    objective.backward()
    np.testing.assert_allclose(loss.detach().cpu().numpy(), arrays["loss"], rtol=1e-5 * amp, atol=1e-5 * amp)
    assert list(q.shape) == meta["q_shape"], (q.shape, meta["q_shape"])
    flat_q = q.detach().cpu().reshape(-1, q.shape[-1]).numpy()
    np.testing.assert_allclose(flat_q[arrays["q_rows"]], arrays["q_vals"], atol=Q_TOL, rtol=0)
    grad = x.grad.detach().cpu()
    gcl = grad if case.get("channel_last", True) else grad.movedim(1, -1)
    gflat = gcl.reshape(-1, gcl.shape[-1]).numpy()
    scale = float(np.abs(arrays["g_vals"]).max()) if "g_full" not in arrays else float(np.abs(arrays["g_full"]).max())
    tol = 1e-5 * amp * max(scale, 1e-12)
    np.testing.assert_allclose(gflat[arrays["g_rows"]], arrays["g_vals"], atol=tol, rtol=1e-4 * amp)
    if "g_full" in arrays:
        np.testing.assert_allclose(grad.numpy(), arrays["g_full"], atol=tol, rtol=1e-4 * amp)
    if "ema_embeddings" in arrays:
        np.testing.assert_allclose(mod._codebook.embeddings.detach().cpu().numpy(), arrays["ema_embeddings"], rtol=1e-4,
                                   atol=1e-5)
    if "cb_grad" in arrays:
        cg = mod._codebook.embeddings.grad.detach().cpu().numpy()
        np.testing.assert_allclose(cg, arrays["cb_grad"], rtol=1e-4, atol=1e-5 * max(1e-12, float(np.abs(arrays["cb_grad"]).max())))
    s = float(grad.double().abs().sum())
    assert abs(s - meta["g_checksum"][1]) <= 1e-4 * amp * max(1e-12, meta["g_checksum"][1]), (s, meta["g_checksum"])
    return float(loss.detach().sum())
