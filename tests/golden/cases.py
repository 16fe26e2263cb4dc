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
    _vq("mh_shared_S", 256, 512, (2, 64, 256), "S", heads=4, codebook_dim=64, separate_codebook_per_head=False),
    _vq("mh_shared_S_train", 256, 512, (2, 64, 256), "S", heads=4, codebook_dim=64,
        separate_codebook_per_head=False, training=True),
    _vq("mh_sep_S_train", 128, 256, (2, 64, 128), "S", heads=2, codebook_dim=64,
        separate_codebook_per_head=True, training=True),
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
    dict(name="grvq", kind="grvq", dim=128, groups=2, Q=3, K=128, x_shape=[2, 64, 128], cls="S", training=False),
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
    # --- cfg5: K=65536, D=512 (M reduced); the sharded search must reproduce the full-codebook idx --
    _vq("cfg5_S", 512, 65536, (1, 64, 512), "S"),
]

CASES_BY_NAME = {c["name"]: c for c in CASES}
