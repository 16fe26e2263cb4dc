/*
 * vq_mi355x.h -- C ABI of the MI355X (gfx950) nearest-codebook library  (libvq_mi355x.so).
 *
 * The reference (MisterBourbaki/vector-quantization-by-ml) has NO native layer: its hot path is a
 * sequence of ATen calls issued from Python.  Each entry point below therefore replaces a run of
 * reference Python lines (cited per function, paths relative to the reference root) rather than an
 * existing FFI symbol, and is what a maintainer would bind (ctypes stub in INTEGRATION.md).
 *
 * Conventions
 *   - all pointers are DEVICE pointers (HBM) unless stated; the library never allocates or frees
 *     persistent memory and never synchronises the host: work is enqueued on `stream`
 *     (a hipStream_t passed as void*; NULL = default stream);
 *   - all floating point is fp32, indices are int64 (reference: codebooks.py:354, general.py:128);
 *   - strides are in ELEMENTS; rows of x / out may be strided (head-split views need no copy);
 *   - return value: 0 on success, a negative VQ_E* code for argument errors, or a positive
 *     hipError_t.  vq_last_error() returns a human-readable string for the calling thread.
 *   - re-entrant across streams; no global mutable state besides the per-thread error string.
 */
#ifndef VQ_MI355X_H
#define VQ_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VQ_METRIC_EUCLID 0 /* similarity = -cdist(x, c)      codebooks.py:128-129 */
#define VQ_METRIC_DOT 1    /* similarity = einsum(x, c)      codebooks.py:122-123 */

#define VQ_E_BADARG (-1)
#define VQ_E_UNSUPPORTED (-2)
#define VQ_E_NODEVICE (-3)

/* flags */
#define VQ_F_STE 1u          /* out = x + (q - x)  (train-mode straight-through, vector_quantize_pytorch.py:273) */
#define VQ_F_FORCE_SIMPLE 2u /* use the scalar-FMA fallback kernel instead of the MFMA kernel (cross-check)        */
#define VQ_F_FORCE_SPLIT 4u  /* force the split-K + packed-key path even when the fused path would be chosen        */
#define VQ_F_X_F16 16u        /* a->x points to fp16 rows (strides in elements); inference only: no STE / sq_err          */
#define VQ_F_X_BF16 32u       /* a->x points to bf16 rows; both are widened in the prologue = the reference's x.float() */
#define VQ_F_SQERR_PER_HEAD 8u /* sq_err (and grad_sq_err of the backward) are [H][Q]: one sum per head and stage     */

/*
 * Packed codebook image (the layout the search kernel streams through LDS).  For one codebook of
 * K codes x D dims:  Kp rows of (Dp + 4) floats, Dp = padded dim chosen by the library (32/64/128/256/512) and Kp = K
 * rounded up to the staged LDS tile (32 codes at Dp >= 256, else 256 / Dp * 32: 64 / 128 / 256 codes at Dp = 128 / 64 / 32);
 * inside each group of 8 dims the even dims come first, then the odd ones, values pre-scaled by -2 (Euclid) or 1 (dot);
 * float Dp of each row holds |c|^2 (d-ordered fmaf chain; +inf for the padding rows), float Dp + 1 holds 1.0, float
 * Dp + 2 the same chain under either metric; the last 16 bytes of the image hold a "some code is non-finite" word.
 * D > 512: ceil(D / 512) such images back to back, one per 512-dim slice of the rows (the last one as wide as it needs).
 * Returns the number of floats ONE packed codebook occupies (including over-copy slack), 0 on bad args.
 */
int64_t vq_packed_floats(int K, int D);

/*
 * Pack `n_codebooks` natural row-major [K, D] codebooks (consecutive ones `cb_stride` floats apart)
 * into `packed` (consecutive images vq_packed_floats(K, D) floats apart).
 * Replaces: the per-call operand preparation inside ATen cdist (cat([-2x,|x|^2,1]) / cat([c,1,|c|^2])),
 * reached from codebooks.py:386.
 */
int vq_pack_codebooks_f32(const float *cb, int n_codebooks, int64_t cb_stride, int K, int D, int metric,
                          float *packed, void *stream);

typedef struct vq_args {
    /* problem */
    int32_t H;      /* independent codebooks searched side by side (heads)  -- grid.y              */
    int32_t Q;      /* residual stages (1 = plain VectorQuantize)                                  */
    int64_t M;      /* rows per head                                                               */
    int32_t K;      /* codes per codebook                                                          */
    int32_t D;      /* dims                                                                        */
    int32_t metric; /* VQ_METRIC_*                                                                 */
    uint32_t flags; /* VQ_F_*                                                                      */
    /* inputs */
    const float *x;        /* [H][M][D], element (h, m, d) at x[h*x_hs + m*x_rs + d]               */
    int64_t x_rs, x_hs;
    const float *cb;       /* natural codebooks: (h, q, k, d) at cb[h*cb_hs + q*cb_qs + k*D + d]    */
    int64_t cb_hs, cb_qs;  /* (cb_qs = 0: all stages share one codebook)                           */
    const float *packed;   /* packed images: codebook (h, q) at packed[h*pk_hs + q*pk_qs]           */
    int64_t pk_hs, pk_qs;
    /* outputs (any of out / best / sq_err may be NULL) */
    float *out;            /* quantized (or straight-through) rows, (h, m, d) at out[h*out_hs + m*out_rs + d];
                              for Q > 1: ((0 + q_1) + q_2) + ...   residual_vq.py:233                */
    int64_t out_rs, out_hs;
    int64_t *idx;          /* (h, m, q) at idx[h*idx_hs + m*idx_rs + q*idx_qs]                      */
    int64_t idx_rs, idx_hs, idx_qs;
    float *best;           /* winning sqrt-distance (Euclid) / similarity (dot); same indexing as idx */
    double *sq_err;        /* [Q] : sum over all heads/rows/dims of (q - x)^2 per stage (overwritten;
                              fixed summation order -> run-to-run reproducible);
                              commitment loss = weight * sq_err / (H*M*D)   vector_quantize_pytorch.py:362 */
    /* scratch */
    void *workspace;       /* >= vq_workspace_bytes() bytes, 16-byte aligned                        */
    int64_t workspace_bytes;
} vq_args;

/* Bytes of scratch vq_quantize_f32 / vq_search_keys_f32 may need for (H, M, Q): packed keys (+ 8 MiB of key planes for K
 * splits), squared-error partials and -- residual stacks (Q > 1) of more than 32 768 rows -- 64 MiB for the residual rows of a
 * partly filled last round of workgroups, which is searched stage by stage (residual_vq.py:212-243 on the remainder). */
int64_t vq_workspace_bytes(int H, int64_t M, int Q);

/*
 * The same for rows WIDER than 512 dims (Q == 1).  The reference has no limit on the row width (cdist / einsum over any
 * D, codebooks.py:122-129,386); here a distance is one k-ordered fmaf chain over all dims, so such rows are swept in
 * 512-dim slices and the chains of one (row chunk) x (code chunk) wait in the workspace between two slices
 * (at most 512 MiB beyond vq_workspace_bytes(H, M, 1)).  For D <= 512 this is vq_workspace_bytes(H, M, 1).
 */
int64_t vq_workspace_bytes_wide(int H, int64_t M, int K, int D);

/*
 * The hot path: search (+ residual loop) + gather + straight-through + squared-error sums.
 * Replaces Codebook.forward's search core  codebooks.py:386-397  (similarity -> first argmax -> gather),
 * VectorQuantize.forward's quantize step   vector_quantize_pytorch.py:261-279,337,361-364  and, for Q > 1,
 * the ResidualVQ loop                      residual_vq.py:154-155,212-243.
 * Bit-exact twin: oracle/vq_oracle.c (k-ordered fmaf chains).
 */
int vq_quantize_f32(const vq_args *a, void *stream);

/*
 * vq_quantize_f32 for Q == 1 that ALSO emits lse[h*M + m] = log sum_k exp(similarity[h, m, k]) from the same sweep
 * (online softmax in the search epilogue).  With a->best (the winner's distance / similarity) this is everything the
 * cross-entropy commitment loss needs -- vector_quantize_pytorch.py:338-346 -- so that loss costs no second sweep:
 * logit[argmax] = -best (Euclid) / best (dot).  D <= 512, fused path only (no VQ_F_FORCE_* flags).
 */
int vq_quantize_lse_f32(const vq_args *a, float *lse, void *stream);

/* Thin named wrappers: Q must be 1 for vq_nearest_f32. */
int vq_nearest_f32(const vq_args *a, void *stream);
int vq_residual_f32(const vq_args *a, void *stream);

/* Largest number of residual stages ONE fused launch can hold for rows of dimension D (the winners' indices and, with
 * want_sq_err, the loss partials of every stage live in the CU's 160 KiB of LDS).  0 for D > 512 (such rows are searched one stage per call, in 512-dim slices).
 * A caller with more stages (the reference's ResidualVQ has no limit, residual_vq.py:212-243) runs its layers one
 * launch each instead.  Host-side arithmetic only: no device call. */
int vq_max_fused_stages(int D, int want_sq_err);

/*
 * Codebook-sharded search, step 1: search only the local shard and emit one packed SIGNED 64-bit key
 * per row:  hi = order image of the value ("smaller wins", top bit flipped), lo = code index + idx_offset.
 * `keys[h*M + m]` is combined with atomic MIN, so the caller initialises it (vq_keys_init) and may then
 * reduce keys across GPUs with ncclMin on int64 (RCCL).  Uses a->x, a->packed, H, M, K, D, metric; Q == 1.
 */
int vq_keys_init(int64_t *keys, int64_t n, void *stream);
int vq_search_keys_f32(const vq_args *a, int64_t idx_offset, int64_t *keys, void *stream);

/*
 * The same search without the init launch and without atomics: when the library splits K over workgroups (few rows: the
 * chip would not be full otherwise), split z STORES its winners into plane z of `keys` [vq_key_planes(a)][H][M]; the
 * winner of a row is the MIN over the planes -- and over the planes of the other shards, which is what the exchange of
 * the sharded path transports and vq_finalize_key_planes_f32 reduces.  vq_key_planes is host arithmetic (same arguments
 * as the search call, current device's CU count).  Rows wider than 512 dims / VQ_F_FORCE_SIMPLE: one plane, initialised
 * and combined inside the call.
 */
int vq_key_planes(const vq_args *a);
int vq_search_key_planes_f32(const vq_args *a, int64_t idx_offset, int64_t *keys, void *stream);

/*
 * Step 2: decode the (reduced) keys and finish: idx/best, gather from the natural codebook `a->cb`
 * (indexed by the GLOBAL code index), straight-through, sq_err.  Q == 1.
 * vq_finalize_key_planes_f32: `keys` holds n_planes candidate planes [n_planes][H][M] (K splits x shards, e.g. the
 * all-gathered planes of every rank); the MIN over the planes is taken on the fly.
 */
int vq_finalize_keys_f32(const vq_args *a, const int64_t *keys, void *stream);
int vq_finalize_key_planes_f32(const vq_args *a, const int64_t *keys, int n_planes, void *stream);

/*
 * Backward of vq_quantize_f32 with respect to x (the autograd of the quantize step, vector_quantize_pytorch.py:261-279,
 * 361-364, and of the residual loop, residual_vq.py:212-243), one pass:
 *   grad_x = (VQ_F_STE ? Q * grad_out : 0) + sum_q 2 * grad_sq_err[q] * (r_q - c_q[idx_q])
 * Uses a->x, a->cb (cb_qs = 0: shared codebook), a->idx as written by the forward, H, M, D, Q, flags.  grad_out (same
 * layout conventions as out) and grad_sq_err ([Q] doubles on the device) may each be NULL.  The codebook receives no
 * gradient here (learnable codebooks take the PyTorch path).
 */
int vq_quantize_backward_f32(const vq_args *a, const float *grad_out, int64_t go_rs, int64_t go_hs, const double *grad_sq_err,
                             float *grad_x, int64_t gx_rs, int64_t gx_hs, void *stream);

/*
 * Training-state step that FOLLOWS the hot path -- exponential-moving-average codebook update.
 * vq_ema_accumulate_f32: counts[h*K + k] += 1 and sums[(h*K + k)*D + d] += x[h, m, d] for every row m assigned to
 *   code k = idx[h*idx_hs + m*idx_rs] (rows with mask[h*M + m] == 0 are skipped; mask may be NULL).  The caller
 *   zeroes counts / sums (and all-reduces them across replicas when codebooks are synchronised).
 *   Replaces embed_onehot.sum(1) and einsum("h n d, h n c -> h c d") -- codebooks.py:405-415 -- without the one-hot.
 * vq_ema_update_f32: cluster_size.lerp_(counts, 1-decay); embed_avg.lerp_(sums, 1-decay); embeddings =
 *   [l2norm](embed_avg / laplace_smoothing(cluster_size) * total) -- codebooks.py:411,417-425.  total_scratch: H floats.
 * The sums are exact up to fp32 summation order (float atomics / per-wave partial sums: run-to-run differences at the
 * 1e-7 relative level).
 */
int vq_ema_accumulate_f32(const float *x, int64_t x_rs, int64_t x_hs, const int64_t *idx, int64_t idx_rs, int64_t idx_hs,
                          const uint8_t *mask, int H, int64_t M, int K, int D, float *counts, float *sums, void *stream);
/* Run-to-run REPRODUCIBLE variant of vq_ema_accumulate_f32 (same arguments, same meaning): no float atomics -- every
 * (row range, code owner) pair stores its partial sums to `workspace` and a second kernel adds the row ranges in a fixed
 * order, so two runs on the same device give bit-identical counts / sums.  workspace: >= vq_ema_det_workspace_bytes()
 * bytes, 16-byte aligned (0 = unsupported: D > 2048).  Meant for the common "many rows per code" regime; with very many
 * codes and few rows it is slower than the atomic path (every owner scans all indices). */
int64_t vq_ema_det_workspace_bytes(int H, int64_t M, int K, int D);
int vq_ema_accumulate_det_f32(const float *x, int64_t x_rs, int64_t x_hs, const int64_t *idx, int64_t idx_rs, int64_t idx_hs,
                              const uint8_t *mask, int H, int64_t M, int K, int D, float *counts, float *sums, void *workspace,
                              int64_t workspace_bytes, void *stream);
/* The same statistics for every stage of a residual stack in one pass (residual_vq.py:212-233: stage q's Codebook sees the
 * residual r_q): uses a->x, a->cb (stages cb_qs apart; 0 = shared), a->idx (as written by vq_quantize_f32), H, M, K, D, Q
 * and VQ_F_STE (train-mode residual rule).  counts [H][Q][K], sums [H][Q][K][D], zeroed by the caller. */
int vq_ema_accumulate_residual_f32(const vq_args *a, float *counts, float *sums, void *stream);
int vq_ema_update_f32(float *cluster_size, float *embed_avg, float *embeddings, const float *counts, const float *sums,
                      float *total_scratch, int H, int K, int D, float decay, float eps, int l2norm, void *stream);

/*
 * Consumers of the similarity matrix.  Both use a->x, a->packed (a->cb for D > 512 / VQ_F_FORCE_SIMPLE),
 * H, M, K, D, metric; Q is ignored.
 * vq_similarities_f32: sims[h*sims_hs + m*sims_rs + k] = -cdist(x, c) (Euclid) or x.c (dot): the third return value of
 *   Codebook.forward -- codebooks.py:386,435 -- bit-identical to the values the search compares.  The caller chooses
 *   how many rows to materialise at once (row chunks via the x / sims pointers).  Rows wider than 512 dims run the sliced
 *   MFMA sweep when a->workspace holds vq_workspace_bytes_wide(H, M, K, D) bytes, else one thread per entry.
 * vq_softmax_stats_f32: logits = scale * similarity; lse[h*M + m] = log sum_k exp(logit), target_logit[h*M + m] = logit
 *   of code target[h*tgt_hs + m*tgt_rs] (0 for a negative = ignored target; -inf for one >= K).  This is
 *   F.cross_entropy(distances, codes, ignore_index=-1) -- vector_quantize_pytorch.py:287-297 -- as an online-softmax
 *   epilogue of the sweep: [M, K] never exists.  target may be NULL (lse only).  Accuracy: native sqrt/exp/log (1e-6 rel).
 */
int vq_similarities_f32(const vq_args *a, float *sims, int64_t sims_rs, int64_t sims_hs, void *stream);
int vq_softmax_stats_f32(const vq_args *a, float scale, const int64_t *target, int64_t tgt_rs, int64_t tgt_hs, float *lse,
                         float *target_logit, void *stream);

/*
 * Backward of that cross entropy with respect to x, fused (flash-attention style, nothing of [M, K] in memory):
 *   grad_x[h, m, :] = coef * d/dx (lse - logit[target])   for rows with target >= 0, 0 for ignored rows
 * = the autograd of F.cross_entropy o (-cdist | einsum) the reference runs -- vector_quantize_pytorch.py:292-294 with
 * ATen's _euclidean_dist_backward.  lse / target_logit: the vq_softmax_stats_f32 outputs (scale 1; the softmax part of
 * the gradient runs through the MFMA sweep, the one-hot part is a rank-one term per row added from a->cb, the natural
 * codebook, with 1 / dist_target taken from target_logit); coef: ONE float on the device
 * (upstream gradient / number of non-ignored rows).  D <= 512 (VQ_E_UNSUPPORTED beyond: use row chunks of
 * vq_similarities_f32).  Gradient with respect to the codebook is not produced.
 */
int vq_ce_backward_f32(const vq_args *a, const float *lse, const float *target_logit, const int64_t *target, int64_t tgt_rs,
                       int64_t tgt_hs, const float *coef, float *grad_x, int64_t gx_rs, int64_t gx_hs, void *stream);

const char *vq_last_error(void);
int vq_device_info(char *buf, size_t n); /* "gfx950 ... CUs" of the current device */

#ifdef __cplusplus
}
#endif
#endif /* VQ_MI355X_H */
