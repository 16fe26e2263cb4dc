"""Golden-case table: one entry per fixture under tests/golden/data/<name>.npz.

Shared by make_golden.py (reference side, this container only) and the tests (build side).
Sizes are the BASELINE configs at reduced M so that the reference and the CPU oracle finish in seconds.
"""
from __future__ import annotations


def _vq(name, dim, K, x_shape, cls="S", **kw):
    d = dict(name=name, kind="vq", dim=dim, K=K, x_shape=list(x_shape), cls=cls, training=False)
    d.update(kw)
    return d


def _rvq(name, dim, Q, K, x_shape, cls="S", **kw):
    d = dict(name=name, kind="rvq", dim=dim, Q=Q, K=K, x_shape=list(x_shape), cls=cls, training=False)
    d.update(kw)
    return d


CASES = [
    # --- cfg1: VectorQuantize dim=64 K=256 on [32,256,64] (full size) -----------------------------
    _vq("cfg1_S", 64, 256, (32, 256, 64), "S"),
    _vq("cfg1_G", 64, 256, (32, 256, 64), "G"),
    _vq("cfg1_Gdup", 64, 256, (32, 256, 64), "Gdup"),
    _vq("cfg1_R", 64, 256, (32, 256, 64), "R"),
    _vq("cfg1_S_train", 64, 256, (32, 256, 64), "S", training=True),
    # --- cfg2: dim=256 K=1024, batch reduced to [4,1024,256] (M=4096) -----------------------------
    _vq("cfg2_S", 256, 1024, (4, 1024, 256), "S"),
    _vq("cfg2_G", 256, 1024, (4, 1024, 256), "G"),
    _vq("cfg2_Gdup", 256, 1024, (4, 1024, 256), "Gdup"),
    _vq("cfg2_R", 256, 1024, (4, 1024, 256), "R"),
    _vq("cfg2_S_train", 256, 1024, (4, 1024, 256), "S", training=True),
    # --- north-star shape K=8192, D=256 (M reduced) ------------------------------------------------
    _vq("k8192_S", 256, 8192, (2, 512, 256), "S"),
    # --- cfg3: multi-head, per-head codebooks, K=8192, dim=512 -> 8 x 64 ---------------------------
    _vq("cfg3a_S", 512, 8192, (2, 128, 512), "S", heads=8, codebook_dim=64, separate_codebook_per_head=True),
    _vq("cfg3a_R", 512, 8192, (2, 128, 512), "R", heads=8, codebook_dim=64, separate_codebook_per_head=True),
    # --- cfg3b: the other reading of BASELINE configs[2]: per-head dim 512, i.e. Linear 512 -> 4096 -> 512
    #     projections around 8 heads x K=8192 x D=512 (M reduced).  The projection weights are seeded on both sides
    #     (seeded_proj: 16 MB of Linear weights do not belong in a fixture); the reference computes them with MKL, the build
    #     with the device GEMM, so only the separated class asks for 100 % equal indices.
    _vq("cfg3b_S", 512, 8192, (1, 64, 512), "S", heads=8, codebook_dim=512, separate_codebook_per_head=True, seeded_proj=True),
    _vq("cfg3b_R", 512, 8192, (1, 64, 512), "R", heads=8, codebook_dim=512, separate_codebook_per_head=True, seeded_proj=True),
    _vq("mh_shared_S", 256, 512, (2, 64, 256), "S", heads=4, codebook_dim=64, separate_codebook_per_head=False),
    _vq("mh_shared_S_train", 256, 512, (2, 64, 256), "S", heads=4, codebook_dim=64,
        separate_codebook_per_head=False, training=True),
    _vq("mh_sep_S_train", 128, 256, (2, 64, 128), "S", heads=2, codebook_dim=64,
        separate_codebook_per_head=True, training=True),
    # --- rows wider than 512 dims (no limit in the reference; here 512-dim slices) -----------------------
    _vq("wide768_S", 768, 300, (2, 64, 768), "S"),
    _vq("wide1100_R", 1100, 512, (1, 96, 1100), "R"),
    _vq("wide640_S_train", 640, 128, (2, 40, 640), "S", training=True),
    _vq("wide_cos_S", 600, 200, (2, 48, 600), "S", use_cosine_sim=True),
    # --- cosine similarity ---------------------------------------------------------------------------
    _vq("cos_S", 64, 512, (4, 128, 64), "S", use_cosine_sim=True),
    _vq("cos_l2_S", 64, 512, (4, 128, 64), "S", use_cosine_sim=True, transform_input="l2norm",
        weights_regularization="l2norm"),
    _vq("cos_G", 64, 512, (4, 128, 64), "G", use_cosine_sim=True),
    # --- layouts -------------------------------------------------------------------------------------
    _vq("chfirst_img", 32, 128, (2, 32, 8, 8), "S", channel_last=False),
    _vq("chfirst_seq", 32, 128, (2, 32, 50), "S", channel_last=False),
    _vq("video", 32, 128, (1, 4, 6, 6, 32), "S"),
    _vq("vec2d", 32, 128, (50, 32), "S"),
    _vq("img_mh", 64, 128, (2, 5, 5, 64), "S", heads=2, codebook_dim=32, separate_codebook_per_head=True),
    # --- projections (codebook_dim != dim): Linear weights are stored in the fixture ----------------
    _vq("proj", 48, 64, (2, 40, 48), "S", codebook_dim=16),
    _vq("proj_mh", 32, 64, (2, 40, 32), "S", heads=2),  # codebook_dim=None & heads>1 -> Linear(32, 64)
    # --- ragged / odd sizes --------------------------------------------------------------------------
    _vq("odd_dims", 100, 300, (3, 37, 100), "S"),
    _vq("odd_small", 5, 7, (2, 33, 5), "S"),
    _vq("k1", 16, 1, (2, 40, 16), "S"),
    _vq("tiny_direct", 8, 16, (1, 10, 8), "S"),  # both sides <= 25 rows: ATen cdist's direct kernel
    # --- masks (variable-length sequences), train mode ----------------------------------------------
    _vq("mask_train", 32, 128, (3, 40, 32), "S", training=True, mask=True),
    _vq("mask_eval", 32, 128, (3, 40, 32), "S", mask=True),
    # --- cfg4: ResidualVQ Q=8 K=1024 dim=256 (M reduced) -------------------------------------------
    _rvq("rvq_S", 256, 8, 1024, (2, 512, 256), "S"),
    _rvq("rvq_S_train", 256, 8, 1024, (2, 512, 256), "S", training=True),
    _rvq("rvq_G", 64, 4, 256, (2, 128, 64), "G"),
    _rvq("rvq_shared", 64, 4, 256, (2, 128, 64), "S", shared_codebook=True),
    _rvq("rvq_allcodes", 32, 3, 64, (2, 20, 32), "S", return_all_codes=True),
    _rvq("rvq_wide_S", 640, 3, 64, (2, 32, 640), "S"),
    dict(name="grvq", kind="grvq", dim=128, groups=2, Q=3, K=128, x_shape=[2, 64, 128], cls="S", training=False),
    _vq("ema_mh_shared_S", 128, 128, (4, 64, 128), "S", training=True, freeze_codebook=False, heads=2, codebook_dim=64,
        cb_extra=dict(threshold_ema_dead_code=0)),
    _vq("ema_chfirst_img_S", 32, 64, (2, 32, 8, 8), "S", training=True, freeze_codebook=False, channel_last=False,
        cb_extra=dict(threshold_ema_dead_code=0)),
    # --- quantize dropout (python's random.Random(seed): reproducible): dropped stages report index -1 and loss 0
    _rvq("rvq_dropout", 64, 6, 64, (2, 40, 64), "S", training=True, return_all_codes=True,
         rvq_extra=dict(quantize_dropout=True, quantize_dropout_cutoff_index=1, quantize_dropout_multiple_of=2),
         fwd_extra=dict(rand_quantize_dropout_fixed_seed=5)),
    _rvq("rvq_dropout_b", 64, 6, 64, (2, 40, 64), "S", training=True,
         rvq_extra=dict(quantize_dropout=True, quantize_dropout_cutoff_index=0, quantize_dropout_multiple_of=1),
         fwd_extra=dict(rand_quantize_dropout_fixed_seed=11)),
    # --- RNG-dependent bookkeeping, pinned on the CPU only (torch.manual_seed before the forward; the sampling consumes the
    #     generator in the reference's order): k-means seeding of the codebook from the first batch, dead-code re-seeding
    _vq("kmeans_init_S", 32, 16, (4, 64, 32), "S", training=True, freeze_codebook=False, cpu_only=True, forward_seed=123,
        cb_extra=dict(initialization_by_kmeans=True, threshold_ema_dead_code=0, kmeans_iter=5)),
    _vq("kmeans_init_cos_S", 32, 16, (4, 64, 32), "S", training=True, freeze_codebook=False, cpu_only=True, forward_seed=123,
        use_cosine_sim=True, transform_input="l2norm", weights_regularization="l2norm",
        cb_extra=dict(initialization_by_kmeans=True, threshold_ema_dead_code=0, kmeans_iter=5)),
    _vq("dead_code_expiry_S", 32, 300, (4, 64, 32), "S", training=True, freeze_codebook=False, cpu_only=True, forward_seed=123,
        cb_extra=dict(threshold_ema_dead_code=2)),
    # cosine k-means on rows that are NOT normalised by the module (the first centroids and the cluster means are
    # raw rows, only the new centroids are normalised, utils/kmeans.py:82-118)
    _vq("kmeans_init_cos_raw_S", 32, 16, (4, 64, 32), "S", training=True, freeze_codebook=False, cpu_only=True, forward_seed=123,
        use_cosine_sim=True, cb_extra=dict(initialization_by_kmeans=True, threshold_ema_dead_code=0, kmeans_iter=5)),
    # heads that SHARE a codebook: re-seeding draws row indices from the reference's "(b h) n" flattening
    _vq("dead_code_expiry_mh_shared_S", 32, 300, (4, 64, 32), "S", training=True, freeze_codebook=False, cpu_only=True,
        forward_seed=123, heads=2, codebook_dim=16, cb_extra=dict(threshold_ema_dead_code=2)),
    _vq("kmeans_init_mh_shared_S", 32, 16, (4, 64, 32), "S", training=True, freeze_codebook=False, cpu_only=True, forward_seed=123,
        heads=2, codebook_dim=16, cb_extra=dict(initialization_by_kmeans=True, threshold_ema_dead_code=0, kmeans_iter=5)),
    dict(name="grvq_train", kind="grvq", dim=128, groups=2, Q=3, K=128, x_shape=[2, 64, 128], cls="S", training=True),
    dict(name="grvq_ema", kind="grvq", dim=128, groups=2, Q=3, K=128, x_shape=[2, 64, 128], cls="S", training=True,
         freeze_codebook=False, cb_extra=dict(threshold_ema_dead_code=0)),
    # --- training-state step after the hot path: EMA update, no dead-code re-seeding (RNG) --
    _vq("ema_S", 64, 256, (8, 256, 64), "S", training=True, freeze_codebook=False, cb_extra=dict(threshold_ema_dead_code=0)),
    _vq("ema_mh_S", 128, 128, (4, 64, 128), "S", training=True, freeze_codebook=False, heads=2, codebook_dim=64,
        separate_codebook_per_head=True, cb_extra=dict(threshold_ema_dead_code=0, decay=0.9)),
    _vq("ema_cos_l2_S", 64, 128, (4, 128, 64), "S", training=True, freeze_codebook=False, use_cosine_sim=True,
        transform_input="l2norm", weights_regularization="l2norm", cb_extra=dict(threshold_ema_dead_code=0)),
    _vq("ema_mask_S", 32, 64, (3, 40, 32), "S", training=True, freeze_codebook=False, mask=True,
        cb_extra=dict(threshold_ema_dead_code=0)),
    _rvq("ema_rvq_S", 64, 3, 128, (2, 128, 64), "S", training=True, freeze_codebook=False,
         cb_extra=dict(threshold_ema_dead_code=0)),
    # --- one codebook shared by all stages AND updated by EMA: every stage searches the codebook the previous one rewrote
    _rvq("ema_rvq_shared_S", 64, 3, 128, (2, 128, 64), "S", training=True, freeze_codebook=False, shared_codebook=True,
         cb_extra=dict(threshold_ema_dead_code=0)),
    # --- residual stack whose layers use the cross-entropy commitment loss (kwargs forwarded to every VectorQuantize)
    _rvq("rvq_ce_train", 64, 3, 128, (2, 128, 64), "S", training=True,
         vq_extra=dict(commitment_use_cross_entropy_loss=True)),
    # --- cfg5: K=65536, D=512 (M reduced); the sharded search must reproduce the full-codebook idx --
    _vq("cfg5_S", 512, 65536, (1, 64, 512), "S"),
]

# --- non-finite inputs: ATen's argmax treats NaN as the maximum and returns the FIRST NaN
#     (utils/general.py:128 over codebooks.py:128-129,386): a row holding a NaN answers 0, a row holding +-inf the first code
#     whose chain meets inf - inf, a code holding a NaN wins every row, a code holding an inf wins the rows on one side of it.
_NF_X = dict(x=[[1, 5, "nan"], [2, 2, "inf"], [3, 2, "-inf"], [4, None, "inf"], [5, 0, "inf"], [5, 1, "-inf"], [6, 63, "nan"],
                [130, 7, "nan"], [255, 0, "-inf"]])
_NF_CB_NAN = dict(cb=[[0, 57, 4, "nan"], [0, 20, 9, "nan"]])
_NF_CB_INF = dict(cb=[[0, 20, 4, "inf"]])
CASES += [
    _vq("nf_x_S", 64, 256, (4, 64, 64), "S", nonfinite=_NF_X),
    _vq("nf_x_S_train", 64, 256, (4, 64, 64), "S", nonfinite=_NF_X, training=True),
    _vq("nf_x_cos_S", 64, 256, (4, 64, 64), "S", nonfinite=_NF_X, use_cosine_sim=True),
    _vq("nf_cb_nan_S", 64, 256, (4, 64, 64), "S", nonfinite=_NF_CB_NAN),
    _vq("nf_cb_nan_cos_S", 64, 256, (4, 64, 64), "S", nonfinite=_NF_CB_NAN, use_cosine_sim=True),
    _vq("nf_cb_inf_S", 64, 256, (4, 64, 64), "S", nonfinite=_NF_CB_INF),
    _vq("nf_cb_inf_cos_S", 64, 256, (4, 64, 64), "S", nonfinite=_NF_CB_INF, use_cosine_sim=True),
    _vq("nf_both_S", 64, 256, (4, 64, 64), "S", nonfinite=dict(x=_NF_X["x"], cb=_NF_CB_INF["cb"])),
    _vq("nf_x_mh_S", 128, 128, (2, 64, 128), "S", heads=2, codebook_dim=64, separate_codebook_per_head=True,
        nonfinite=dict(x=[[1, 5, "nan"], [2, 70, "inf"], [3, 64, "-inf"], [100, 127, "nan"]])),
    _vq("nf_x_d256_S", 256, 1024, (2, 256, 256), "S",
        nonfinite=dict(x=[[0, 255, "nan"], [2, 100, "inf"], [3, 101, "-inf"], [300, 0, "nan"], [511, 17, "inf"]])),
    _vq("nf_x_d512_S", 512, 300, (2, 40, 512), "S", nonfinite=dict(x=[[1, 300, "nan"], [2, 2, "inf"], [79, 511, "-inf"]])),
    _vq("nf_x_wide_S", 768, 300, (2, 40, 768), "S", nonfinite=dict(x=[[1, 700, "nan"], [2, 2, "inf"], [3, 600, "-inf"], [79, 5, "nan"]])),
    _vq("nf_cb_wide_S", 768, 300, (2, 40, 768), "S", nonfinite=dict(cb=[[0, 33, 700, "nan"]])),
    _rvq("nf_rvq_x_S", 64, 3, 128, (2, 64, 64), "S", nonfinite=dict(x=[[1, 5, "nan"], [2, 2, "inf"], [3, 2, "-inf"], [127, 63, "nan"]])),
    _rvq("nf_rvq_x_S_train", 64, 3, 128, (2, 64, 64), "S", training=True,
         nonfinite=dict(x=[[1, 5, "nan"], [2, 2, "inf"], [3, 2, "-inf"], [127, 63, "nan"]])),
    _rvq("nf_rvq_cb_S", 64, 3, 128, (2, 64, 64), "S", nonfinite=dict(cb=[[1, 40, 3, "nan"]])),
    _rvq("nf_rvq_cb_inf_S", 64, 3, 128, (2, 64, 64), "S", nonfinite=dict(cb=[[2, 40, 3, "inf"], [0, 7, 0, "-inf"]])),
]

CASES_BY_NAME = {c["name"]: c for c in CASES}


# ---------------------------------------------------------------------------------------------------------------
# Consumers of the similarity matrix: cross-entropy commitment loss, cross entropy against given
# indices, codebook diversity loss.  Goldens hold the loss, the returned tensors and d loss / d x from the imported
# reference (make_golden.py: run_vqloss).  All in train mode with a frozen codebook unless stated.
# ---------------------------------------------------------------------------------------------------------------
def _vql(name, dim, K, x_shape, **kw):
    d = dict(name=name, kind="vqloss", dim=dim, K=K, x_shape=list(x_shape), cls="S", training=True)
    d.update(kw)
    return d


_CE = dict(commitment_use_cross_entropy_loss=True)

LOSS_CASES = [
    _vql("ce_commit", 64, 256, (4, 64, 64), vq_extra=_CE),
    _vql("ce_commit_mask", 64, 256, (3, 40, 64), vq_extra=_CE, mask=True),
    _vql("ce_commit_mh_sep", 64, 128, (2, 50, 64), vq_extra=_CE, heads=2, codebook_dim=32, separate_codebook_per_head=True),
    _vql("ce_commit_mh_shared", 64, 128, (2, 50, 64), vq_extra=_CE, heads=2, codebook_dim=32),
    _vql("ce_commit_mh_mask", 64, 128, (3, 20, 64), vq_extra=_CE, heads=2, codebook_dim=32, separate_codebook_per_head=True,
         mask=True),
    _vql("ce_commit_cos", 64, 256, (4, 64, 64), vq_extra=_CE, use_cosine_sim=True, transform_input="l2norm",
         weights_regularization="l2norm"),
    _vql("ce_commit_cfg2", 256, 1024, (2, 512, 256), vq_extra=_CE),
    _vql("ce_commit_odd", 100, 301, (3, 37, 100), vq_extra=_CE),
    _vql("ce_commit_w", 64, 256, (4, 64, 64), vq_extra=dict(commitment_use_cross_entropy_loss=True, commitment_weight=0.25)),
    _vql("ce_indices", 64, 256, (4, 64, 64), given_indices=True),
    _vql("ce_indices_eval", 64, 256, (4, 64, 64), given_indices=True, training=False),
    _vql("ce_indices_ignore", 64, 256, (4, 64, 64), given_indices=True, ignore_some=True),
    _vql("ce_indices_mh_sep", 64, 128, (2, 50, 64), given_indices=True, heads=2, codebook_dim=32,
         separate_codebook_per_head=True),
    _vql("ce_indices_mh_shared", 64, 128, (2, 50, 64), given_indices=True, heads=2, codebook_dim=32),
    _vql("ce_indices_img", 32, 64, (2, 32, 6, 6), given_indices=True, channel_last=False),
    _vql("div", 64, 256, (4, 64, 64), vq_extra=dict(codebook_diversity_loss_weight=0.5)),
    _vql("div_t1", 64, 256, (4, 64, 64), vq_extra=dict(codebook_diversity_loss_weight=0.5, codebook_diversity_temperature=1.0)),
    _vql("div_mh_shared", 64, 128, (2, 50, 64), heads=2, codebook_dim=32,
         vq_extra=dict(codebook_diversity_loss_weight=1.0, codebook_diversity_temperature=2.0)),
    _vql("div_mh_sep", 64, 128, (2, 50, 64), heads=2, codebook_dim=32, separate_codebook_per_head=True,
         vq_extra=dict(codebook_diversity_loss_weight=1.0, codebook_diversity_temperature=2.0)),
    _vql("div_cos", 64, 256, (4, 64, 64), use_cosine_sim=True, transform_input="l2norm", weights_regularization="l2norm",
         vq_extra=dict(codebook_diversity_loss_weight=0.5, codebook_diversity_temperature=10.0)),
    _vql("div_ce", 64, 256, (4, 64, 64), vq_extra=dict(codebook_diversity_loss_weight=0.5, codebook_diversity_temperature=1.0,
                                                       commitment_use_cross_entropy_loss=True)),
    # EMA update running (codebook not frozen): losses are evaluated against the pre-update codebook
    _vql("ce_commit_ema", 64, 256, (4, 64, 64), vq_extra=_CE, freeze_codebook=False,
         cb_extra=dict(threshold_ema_dead_code=0)),
]

_LEARN = dict(learnable_codebook=True, ema_update=False)

LOSS_CASES += [
    # the standard training step with the EMA update running: the backward of the MSE commitment loss must use the codes the
    # forward searched, not the ones the EMA step wrote afterwards
    _vql("mse_commit_ema", 64, 256, (4, 64, 64), freeze_codebook=False, backprop_q=True, cb_extra=dict(threshold_ema_dead_code=0)),
    _vql("mse_commit_ema_mh", 64, 128, (2, 50, 64), freeze_codebook=False, backprop_q=True, heads=2, codebook_dim=32,
         separate_codebook_per_head=True, cb_extra=dict(threshold_ema_dead_code=0)),
]

LOSS_CASES += [
    # --- learnable codebook options of the quantize step: gradient scaling rule and the in-place optimizer ------------
    _vql("learnable", 64, 256, (4, 64, 64), cb_extra=_LEARN, freeze_codebook=False, backprop_q=True),
    _vql("sync_update", 64, 256, (4, 64, 64), cb_extra=_LEARN, freeze_codebook=False, backprop_q=True,
         vq_extra=dict(sync_update_v=0.5)),
    _vql("inplace_sgd", 64, 256, (4, 64, 64), cb_extra=_LEARN, freeze_codebook=False, backprop_q=True, inplace_sgd_lr=50.0),
    _vql("inplace_sgd_mask", 32, 64, (3, 40, 32), cb_extra=_LEARN, freeze_codebook=False, mask=True, inplace_sgd_lr=20.0),
    _vql("inplace_sgd_mh", 64, 128, (2, 50, 64), cb_extra=_LEARN, freeze_codebook=False, heads=2, codebook_dim=32,
         separate_codebook_per_head=True, inplace_sgd_lr=50.0),
]

LOSS_CASES_BY_NAME = {c["name"]: c for c in LOSS_CASES}
