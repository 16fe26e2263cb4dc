// vq_kernels.hip -- nearest-codebook search for MI355X (gfx950 / CDNA4).  Hand-written HIP, no
// compatibility layers.  See DESIGN.md for the full description; summary of the data path:
//
//   pack      natural codebook [K, D]  ->  packed image  [Kp][Dp + 4]  (even/odd de-interleave inside
//             each group of 8 dims, pre-scaled by -2 for Euclid, |c|^2 in float Dp of every row)
//   search    one wave owns 32 rows of x for the whole sweep; their fp32 values live in REGISTERS as
//             v_mfma_f32_32x32x2_f32 B-fragments (Dp/2 VGPRs per lane).  The workgroup streams ~33 KB
//             tiles of the packed image HBM/L2 -> LDS with buffer_load ... lds (LDS-DMA, double
//             buffered) and each wave runs Dp/2 MFMAs per 32-code sub-tile (codes on the MFMA i axis,
//             rows on the j axis), then ONE more MFMA that adds |x|^2 * 1 + 1 * |c|^2  (the two
//             augmented GEMM columns of ATen's cdist).  The 32x32 result is reduced in-lane (a lane holds
//             16 codes of ONE row): min3 tree per sub-tile, the record sub-tile's values are parked and
//             the tie-exact rule (correctly rounded sqrt, lowest index) is resolved once per sweep.
//   finalize  gather codebook[idx] (natural layout), straight-through, squared-error sums; fused in
//             the search kernel unless the sweep was split over K (packed 64-bit keys + atomic min).
//
// Arithmetic contract (bit-exact twin: oracle/vq_oracle.c): every distance is the k-ordered fmaf
// chain  fma(1*|c|^2 .. fma(|x|^2*1, fma(x_{D-1}, -2c_{D-1}, ... fma(x_0, -2c_0, 0))))  which is what
// v_mfma_f32_32x32x2_f32 computes; norms are d-ordered fmaf chains; sqrt is correctly rounded.
//
// Reference lines replaced (relative to the reference root): vector_quantization/codebooks.py:386-397,
// utils/general.py:126-136,159-163, vector_quantize_pytorch.py:261-279,361-364, residual_vq.py:212-243.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/vq_mi355x.h"

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int kTileCodes = 32;
constexpr int kPackSlack = 2048;  // floats of over-copy slack behind every packed image
constexpr int kModeFused = 0;
constexpr int kModeKeys = 1;

thread_local char g_err[512] = "";

int fail(int code, const char *msg) {
    snprintf(g_err, sizeof(g_err), "%s", msg);
    return code;
}

int hip_fail(hipError_t e, const char *what) {
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
    return (int)e;
}

inline int padded_dim(int D) {
    if (D <= 32) return 32;
    if (D <= 64) return 64;
    if (D <= 128) return 128;
    if (D <= 256) return 256;
    if (D <= 512) return 512;
    return 0;
}

inline int round_up(int a, int b) { return (a + b - 1) / b * b; }

// A staged LDS tile always carries ~33 KB: 32 codes at Dp >= 256, 64 / 128 / 256 codes at Dp = 128 / 64 / 32, so
// the per-tile barrier and LDS-DMA issue are amortised over the same number of MFMAs at every dim.
constexpr int sub_tiles(int DP) { return DP >= 256 ? 1 : 256 / DP; }

// ------------------------------------------------------------------------------------------------
// packed (value, index) keys: signed 64-bit, MIN wins, lowest index on equal values
// ------------------------------------------------------------------------------------------------
template <int METRIC>
__device__ __forceinline__ long long make_key(float v, long long idx) {
    unsigned b = __float_as_uint(v);
    unsigned m;
    if (METRIC == VQ_METRIC_DOT) {
        unsigned mono = (b & 0x80000000u) ? ~b : (b | 0x80000000u);
        m = ~mono;
    } else {
        m = b;  // sqrt distance >= 0: IEEE bits are order preserving
    }
    return (long long)(((unsigned long long)(m ^ 0x80000000u) << 32) | (unsigned long long)(unsigned)idx);
}

__device__ __forceinline__ float key_value(long long key, int metric) {
    unsigned m = (unsigned)((unsigned long long)key >> 32) ^ 0x80000000u;
    unsigned b;
    if (metric == VQ_METRIC_DOT) {
        unsigned mono = ~m;
        b = (mono & 0x80000000u) ? (mono ^ 0x80000000u) : ~mono;
    } else {
        b = m;
    }
    return __uint_as_float(b);
}

// ------------------------------------------------------------------------------------------------
// pack kernel: one thread per packed row
// ------------------------------------------------------------------------------------------------
__global__ void vq_pack_kernel(const float *__restrict__ cb, long long cb_stride, int K, int Kp, int D, int DP,
                               int metric, float *__restrict__ packed, long long pk_stride) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    const int RS = DP + 4;
    float *img = packed + (long long)blockIdx.y * pk_stride;
    if (k >= Kp) {
        // the threads past the last row zero the over-copy slack behind the image (read by the tile DMA)
        const int nslack = kPackSlack / 4;
        const int j = k - Kp;
        const int nthreads = gridDim.x * blockDim.x - Kp;
        for (int i = j; i < nslack; i += nthreads) *(f32x4 *)(img + (long long)Kp * RS + 4 * i) = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
        return;
    }
    const float *src = cb + (long long)blockIdx.y * cb_stride + (long long)k * D;
    float *dst = img + (long long)k * RS;
    const float scale = (metric == VQ_METRIC_EUCLID) ? -2.0f : 1.0f;
    const bool vec = (D % 4 == 0) && (cb_stride % 4 == 0) && (((uintptr_t)cb & 15) == 0);
    float cn = 0.0f;
#pragma unroll 4
    for (int g = 0; g < DP / 8; ++g) {
        float v[8];
        if (vec && k < K && 8 * g + 8 <= D) {
            const f32x4 lo = *(const f32x4 *)(src + 8 * g), hi = *(const f32x4 *)(src + 8 * g + 4);
            v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w;
            v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int d = 8 * g + e;
                v[e] = (k < K && d < D) ? src[d] : 0.0f;
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) cn = fmaf(v[e], v[e], cn);  // d-ordered chain; padded zeros leave it unchanged
        f32x4 ev = {scale * v[0], scale * v[2], scale * v[4], scale * v[6]};
        f32x4 od = {scale * v[1], scale * v[3], scale * v[5], scale * v[7]};
        *(f32x4 *)(dst + 8 * g) = ev;
        *(f32x4 *)(dst + 8 * g + 4) = od;
    }
    if (k >= K) cn = (metric == VQ_METRIC_EUCLID) ? __builtin_inff() : 0.0f;
    f32x4 tail = {(metric == VQ_METRIC_EUCLID) ? cn : 0.0f, 0.0f, 0.0f, 0.0f};
    *(f32x4 *)(dst + DP) = tail;
}

// ------------------------------------------------------------------------------------------------
// search kernel
// ------------------------------------------------------------------------------------------------
struct SearchParams {
    const float *x;
    long long x_rs, x_hs;
    const float *cb;
    long long cb_hs, cb_qs;
    const float *packed;
    long long pk_hs, pk_qs;
    float *out;
    long long out_rs, out_hs;
    long long *idx;
    long long idx_rs, idx_hs, idx_qs;
    float *best;
    float *loss_part;  // [H * gridDim.x * WAVES][Q] or NULL
    long long *keys;   // kModeKeys
    long long idx_offset;
    long long M;
    int K, D, Q;
    int ntiles, tiles_per_split;
    unsigned pk_bytes;  // bytes of one packed image (buffer descriptor range)
    int mode;
    int ste;
    int vec_x;    // x rows may be read as float4 (D % 4 == 0, strides % 4 == 0, 16-B aligned base)
    int vec_fin;  // finalize may use float4 on x / out / cb
};

typedef __attribute__((address_space(3))) f32x4 lds_f32x4;


#ifdef VQ_EXP_STAMPS
__device__ unsigned long long g_stamps[8192 * 4];
#define STAMP(i) do { if (lane == 0) g_stamps[(((long long)blockIdx.x * WAVES + wave) & 8191) * 4 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define STAMP(i) do { } while (0)
#endif

// buffer_load_dwordx4 ... lds: `base` + voffset (per lane) + soffset (scalar) -> LDS at l + lane * 16
__device__ __forceinline__ void lds_dma16(const float *base, unsigned bytes, int voffset, int soffset, lds_f32x4 *l) {
#if defined(__HIP_DEVICE_COMPILE__)
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)base, (short)0, (int)bytes, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)l, 16, voffset, soffset, 0, 0);
#else
    (void)base; (void)bytes; (void)voffset; (void)soffset; (void)l;
#endif
}

template <int DP, int WAVES>
struct Geo {
    static constexpr int RS = DP + 4;                       // packed row stride, floats
    static constexpr int RS4 = RS / 4;
    static constexpr int SUB = sub_tiles(DP);               // 32-code MFMA sub-tiles per staged tile / barrier
    static constexpr int TILE_CODES = kTileCodes * SUB;
    static constexpr int TILE_F4 = TILE_CODES * RS / 4;     // float4 per staged tile image
    static constexpr int TILE_CHUNKS = (TILE_F4 + 63) / 64; // 1-KiB wave copies per tile (over-copy)
    static constexpr int BUF_F4 = TILE_CHUNKS * 64;
    static constexpr int CH = DP < 64 ? DP : 64;            // prologue column chunk
    static constexpr int XS = CH + 4;                       // prologue scratch row stride, floats (16-B rows)
    static constexpr int NS = DP / 2;                       // MFMA k-steps
    static constexpr int NBUF = 2;
    static constexpr int MAIN_FLOATS = (NBUF * BUF_F4 * 4 > WAVES * 32 * XS) ? NBUF * BUF_F4 * 4 : WAVES * 32 * XS;
    static constexpr int NCH4 = DP >= 256 ? DP / 256 : 1;   // float4 chunks per lane in finalize
    static constexpr int NEL = DP >= 64 ? DP / 64 : 1;      // scalars per lane in finalize
};

// code fragments: one ds_read_b128 feeds 4 MFMAs (256 cycles of matrix pipe).  `a` is the whole tile's fragment
// array (compile-time indexed -> registers); reads run PF groups ahead of the MFMAs and the order is pinned so the
// scheduler cannot hoist every read to the top (register pressure).  Groups [G0, G1) of 8 dims.
template <int DP>
struct FragPipe {
    static constexpr int NG = DP / 8;
    static constexpr int PF = NG < 3 ? NG : 3;
};

template <int DP>
__device__ __forceinline__ void mfma_prefetch(f32x4 (&a)[DP / 8], const f32x4 *ta) {
#pragma unroll
    for (int g = 0; g < FragPipe<DP>::PF; ++g) a[g] = ta[2 * g];
}

template <int DP, int G0, int G1>
__device__ __forceinline__ void mfma_range(f32x16 &acc, f32x4 (&a)[DP / 8], const f32x4 *ta, const float (&xf)[DP / 2]) {
    constexpr int NG = FragPipe<DP>::NG, PF = FragPipe<DP>::PF;
#pragma unroll
    for (int g = G0; g < G1; ++g) {
        if (g + PF < NG) a[g + PF] = ta[2 * (g + PF)];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g].x, xf[4 * g + 0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g].y, xf[4 * g + 1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g].z, xf[4 * g + 2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g].w, xf[4 * g + 3], acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// End of a 32-code tile for one wave: augmented-column MFMA (Euclid), tail masking, in-lane reduction.
// A lane holds 16 codes of ONE row: acc[r] <-> code t*32 + 4*h + (r&3) + 8*(r>>2), ascending in r.
__device__ __forceinline__ float vmin3(float a, float b, float c) {
    float o;  // raw instruction: no canonicalising v_max in front (MFMA outputs are already canonical)
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(o) : "v"(a), "v"(b), "v"(c));
    return o;
}
__device__ __forceinline__ float vmax3(float a, float b, float c) {
    float o;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(o) : "v"(a), "v"(b), "v"(c));
    return o;
}

// Per-lane running state of the argmin over the codes a lane has seen (16 per 32-code sub-tile, ascending).
// `pend` parks the 16 values of the sub-tile that produced the current record low; the expensive part of the
// tie-exact rule (correctly rounded sqrt, rounding-band threshold, lowest index inside the band) is resolved ONCE at
// the end of the sweep instead of on every record (with 64 lanes x few tiles some lane sets a record on nearly every
// tile, and on gfx950 those vector instructions are not hidden behind the f32 MFMA stream).
struct LaneBest {
    float best_t;   // Euclid: min clamped squared distance so far; dot: max similarity so far
    int pend_u;     // sub-tile that produced it
    f32x16 pend;    // its 16 values
};

__device__ __forceinline__ float rounding_band_hi(float sq) {
    // largest fp32 t whose correctly rounded sqrt is still `sq`:  t < (sq + ulp(sq)/2)^2, exact in fp64
    const float up = __uint_as_float(__float_as_uint(sq) + 1u);  // next float above (sq >= 0)
    const double mid = (double)sq + 0.5 * ((double)up - (double)sq);
    const double lim = mid * mid;
    float hi = (float)lim;
    if ((double)hi >= lim) hi = __uint_as_float(__float_as_uint(hi) - 1u);
    return hi;
}

// End of a 32-code sub-tile for one wave: tail masking + in-lane reduction of the 32x32 result.
// A lane holds 16 codes of ONE row: acc[r] <-> code u*32 + 4*h + (r&3) + 8*(r>>2), ascending in r.
template <int METRIC, int DP>
__device__ __forceinline__ void tile_epilogue(f32x16 &acc, int u, int h, int K, LaneBest &st) {
    const float INF = __builtin_inff();
    const int cbase = u * kTileCodes + 4 * h;
    if (METRIC == VQ_METRIC_EUCLID) {
        if (u * kTileCodes + kTileCodes > K) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (cbase + (r & 3) + 8 * (r >> 2) >= K) acc[r] = INF;
        }
        float tm = vmin3(acc[0], acc[1], acc[2]);
        tm = vmin3(tm, acc[3], acc[4]);
        tm = vmin3(tm, acc[5], acc[6]);
        tm = vmin3(tm, acc[7], acc[8]);
        tm = vmin3(tm, acc[9], acc[10]);
        tm = vmin3(tm, acc[11], acc[12]);
        tm = vmin3(tm, acc[13], acc[14]);
        tm = vmax3(fminf(tm, acc[15]), 0.0f, 0.0f);  // clamp_min_(0) commutes with min
        if (tm < st.best_t) {
            // A new record low of the squared distance.  It replaces the parked sub-tile unless its sqrt ROUNDS to the
            // same value as the parked minimum's (then the earlier sub-tile keeps the win: lower codes).  Two values
            // more than 2^-21 apart (relative) cannot share a rounded sqrt, so only near-ties pay for the two sqrts.
            bool take = true;
            if (tm * 1.0000004768371582f >= st.best_t) take = sqrtf(tm) < sqrtf(st.best_t);
            if (take) {
                st.best_t = tm;
                st.pend_u = u;
                st.pend = acc;
            }
        }
    } else {
        if (u * kTileCodes + kTileCodes > K) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (cbase + (r & 3) + 8 * (r >> 2) >= K) acc[r] = -INF;
        }
        float tm = vmax3(acc[0], acc[1], acc[2]);
        tm = vmax3(tm, acc[3], acc[4]);
        tm = vmax3(tm, acc[5], acc[6]);
        tm = vmax3(tm, acc[7], acc[8]);
        tm = vmax3(tm, acc[9], acc[10]);
        tm = vmax3(tm, acc[11], acc[12]);
        tm = vmax3(tm, acc[13], acc[14]);
        tm = vmax3(tm, acc[15], acc[15]);
        if (tm > st.best_t) {  // strictly better than everything earlier
            st.best_t = tm;
            st.pend_u = u;
            st.pend = acc;
        }
    }
}

// End of a sweep: turn the parked sub-tile into (value in the compared space, lowest winning code of this lane).
template <int METRIC>
__device__ __forceinline__ void resolve_best(const LaneBest &st, int h, float &best_s, int &best_i) {
    int bi = 0;
    if (METRIC == VQ_METRIC_EUCLID) {
        best_s = sqrtf(st.best_t);  // correctly rounded; +inf if the lane saw no finite distance
        const float hi = (st.best_t < __builtin_inff()) ? rounding_band_hi(best_s) : __builtin_inff();
#pragma unroll
        for (int r = 15; r >= 0; --r) bi = (st.pend[r] <= hi) ? (r & 3) + 8 * (r >> 2) : bi;  // unclamped: t < 0 -> 0 <= hi
    } else {
        best_s = st.best_t;
#pragma unroll
        for (int r = 15; r >= 0; --r) bi = (st.pend[r] == st.best_t) ? (r & 3) + 8 * (r >> 2) : bi;
    }
    best_i = st.pend_u * kTileCodes + 4 * h + bi;
}

template <int DP, int WAVES, int METRIC, bool MULTI>
__global__ void __launch_bounds__(WAVES * 64, (DP <= 256 ? 2 : 1)) vq_search_mfma(const SearchParams p) {
    using G = Geo<DP, WAVES>;
    constexpr int RS = G::RS, RS4 = G::RS4, CH = G::CH, XS = G::XS, NS = G::NS;
    constexpr bool EUCLID = (METRIC == VQ_METRIC_EUCLID);

    extern __shared__ __attribute__((aligned(16))) float smem[];
    f32x4 *tile4 = (f32x4 *)smem;
    lds_f32x4 *tile4_lds = (lds_f32x4 *)smem;  // same bytes, LDS address space (LDS-DMA destinations)
    int *sidx = (int *)(smem + G::MAIN_FLOATS);  // [WAVES][Q][32]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 31, h = lane >> 5;
    const int head = blockIdx.y;
    const long long row0 = ((long long)blockIdx.x * WAVES + wave) * 32;
    const float *xh = p.x + (long long)head * p.x_hs;
    const float INF = __builtin_inff();

    // ---------------- prologue: this wave's 32 rows -> MFMA fragments in registers ----------------
    // xf[s] = x[row0 + c][2 s + h]   (lane half h holds the k = h operand of MFMA step s)
    STAMP(0);
    // Memory-bound phases (prologue, finalize) run at raised priority: when another workgroup's wave is streaming
    // MFMAs on the same SIMD, these few load/store/LDS instructions must not be starved by it.
    __builtin_amdgcn_s_setprio(2);
    float xf[NS];
    float xn0 = 0.0f;  // |x|^2 of row c: d-ordered fmaf chain (the oracle's sumsq_chain)
    {
        // Wave-private staging region, no workgroup barriers: a wave's LDS operations execute in order.
        // Global loads of chunk i+1 are in flight while chunk i goes through LDS.
        float *xs = smem + wave * (32 * XS);
        constexpr int NCHUNK = DP / CH;
        constexpr int LPL = CH / 8;  // float4 loads per lane per chunk
        f32x4 v[2][LPL];
        auto load_chunk = [&](int ch, f32x4 (&dst)[LPL]) {
#pragma unroll
            for (int it = 0; it < LPL; ++it) {
                const int f = it * 64 + lane;
                const int r = f / (CH / 4), c4 = f % (CH / 4);
                long long grow = row0 + r;
                if (grow >= p.M) grow = p.M - 1;
                const int d0 = ch * CH + c4 * 4;
                const float *src = xh + grow * p.x_rs + d0;
                f32x4 t = {0.0f, 0.0f, 0.0f, 0.0f};
                if (p.vec_x) {
                    if (d0 < p.D) t = __builtin_nontemporal_load((const f32x4 *)src);  // each row is read once
                } else {
                    if (d0 + 0 < p.D) t.x = src[0];
                    if (d0 + 1 < p.D) t.y = src[1];
                    if (d0 + 2 < p.D) t.z = src[2];
                    if (d0 + 3 < p.D) t.w = src[3];
                }
                dst[it] = t;
            }
        };
        load_chunk(0, v[0]);
#pragma unroll
        for (int ch = 0; ch < NCHUNK; ++ch) {
            if (ch + 1 < NCHUNK) load_chunk(ch + 1, v[(ch + 1) & 1]);
#pragma unroll
            for (int it = 0; it < LPL; ++it) {
                const int f = it * 64 + lane;
                const int r = f / (CH / 4), c4 = f % (CH / 4);
                *(f32x4 *)(xs + r * XS + c4 * 4) = v[ch & 1][it];  // XS = CH + 4: b128 accesses conflict-free
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const float *rp = xs + c * XS;  // this lane's row (both lane halves read the same row)
#pragma unroll
            for (int j = 0; j < CH / 8; ++j) {
                const f32x4 lo = *(const f32x4 *)(rp + 8 * j);
                const f32x4 hi = *(const f32x4 *)(rp + 8 * j + 4);
                if (EUCLID) {
                    xn0 = fmaf(lo.x, lo.x, xn0);
                    xn0 = fmaf(lo.y, lo.y, xn0);
                    xn0 = fmaf(lo.z, lo.z, xn0);
                    xn0 = fmaf(lo.w, lo.w, xn0);
                    xn0 = fmaf(hi.x, hi.x, xn0);
                    xn0 = fmaf(hi.y, hi.y, xn0);
                    xn0 = fmaf(hi.z, hi.z, xn0);
                    xn0 = fmaf(hi.w, hi.w, xn0);
                    asm volatile("" : "+v"(xn0));  // pin the chain here: do not keep lo/hi alive to finish it later
                }
                // lower half-wave keeps dims 8j..8j+3, upper 8j+4..8j+7; two half-wave exchanges de-interleave
                // them into the MFMA k-parity layout: lower gets the even dims, upper the odd dims.
                const f32x4 m = h ? hi : lo;
                const auto xy = __builtin_amdgcn_permlane32_swap(__float_as_uint(m.x), __float_as_uint(m.y), false, false);
                const auto zw = __builtin_amdgcn_permlane32_swap(__float_as_uint(m.z), __float_as_uint(m.w), false, false);
                const int sb = ch * (CH / 2) + 4 * j;
                xf[sb + 0] = __uint_as_float(xy[0]);  // dim 8j + 0 + h
                xf[sb + 1] = __uint_as_float(zw[0]);  // dim 8j + 2 + h
                xf[sb + 2] = __uint_as_float(xy[1]);  // dim 8j + 4 + h
                xf[sb + 3] = __uint_as_float(zw[1]);  // dim 8j + 6 + h
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // reads done before the next chunk overwrites
        }
        __syncthreads();  // the staging region is about to be reused as codebook tile buffers
    }

    STAMP(1);
    __builtin_amdgcn_s_setprio(0);
    const long long row = row0 + c;
    const bool row_ok = row < p.M;

    for (int q = 0; q < (MULTI ? p.Q : 1); ++q) {
        const float *pk = p.packed + (long long)head * p.pk_hs + (long long)q * p.pk_qs;

        // |x|^2 of the current residual: stage 0 has it from the prologue; later stages take the diagonal of
        // X X^T (a k-ordered fmaf chain on the matrix pipe, bit-identical to the oracle's sumsq_chain)
        float b_aug = 1.0f;
        if (EUCLID && q == 0) {
            b_aug = h ? 1.0f : xn0;  // B[k=0][row] = |x|^2, B[k=1][row] = 1
        } else if (EUCLID) {
            f32x16 d = {0};
#pragma unroll
            for (int s = 0; s < NS; ++s) d = __builtin_amdgcn_mfma_f32_32x32x2f32(xf[s], xf[s], d, 0, 0, 0);
            const int rsel = (c & 3) + 4 * (c >> 3);
            float dv = 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) dv = (r == rsel) ? d[r] : dv;
            const float ov = __shfl_xor(dv, 32);
            const float xn = (h == ((c >> 2) & 1)) ? dv : ov;
            b_aug = h ? 1.0f : xn;  // B[k=0][row] = |x|^2, B[k=1][row] = 1
        }

        LaneBest lb;
        lb.best_t = EUCLID ? INF : -INF;
        lb.pend_u = 0;
        lb.pend = (f32x16){0};
        if (EUCLID) {
#pragma unroll
            for (int r = 0; r < 16; ++r) lb.pend[r] = INF;  // nothing parked yet (pend[0] <= +inf still selects code 0)
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) lb.pend[r] = -INF;
        }

        const int t0 = blockIdx.z * p.tiles_per_split;
        const int t1 = (t0 + p.tiles_per_split < p.ntiles) ? t0 + p.tiles_per_split : p.ntiles;

        // LDS-DMA of one tile image: buffer_load ... lds with the per-lane part (lane * 16 B) in voffset and the
        // tile / chunk position in the SCALAR offset -> no vector instructions at all per issue (on gfx950 VALU
        // work is not free beside f32 MFMA: it executes on the same lanes).
        auto stage = [&](int tile, int buf) {
#pragma unroll
            for (int i = 0; i < (G::TILE_CHUNKS + WAVES - 1) / WAVES; ++i) {
                const int ck = i * WAVES + wave;
                if (ck < G::TILE_CHUNKS)
                    lds_dma16(pk, p.pk_bytes, lane * 16, (tile * G::TILE_F4 + ck * 64) * 16,
                              tile4_lds + buf * G::BUF_F4 + ck * 64);
            }
        };
        // One 32-code sub-tile: Dp/2 MFMAs into `acc`, then the augmented-column MFMA (|x|^2 * 1 + 1 * |c|^2).
        // Software pipelined inside the wave: the PREVIOUS sub-tile's reduction (`prev`, finished long ago, so no
        // MFMA drain) and the NEXT tile's LDS-DMA issue are expanded between MFMA groups, where they issue while
        // the matrix pipe is busy with this sub-tile.  `u` counts 32-code sub-tiles from code 0.
        constexpr int SUB = G::SUB;
        // One 32-code sub-tile: Dp/2 MFMAs into `acc`, then the augmented-column MFMA (|x|^2 * 1 + 1 * |c|^2).
        // Software pipelined inside the wave: the PREVIOUS sub-tile's reduction (`prev`, finished long ago, so no
        // MFMA drain) and the NEXT tile's LDS-DMA issue are expanded between MFMA groups.  `u` counts 32-code
        // sub-tiles from code 0.  (Measured alternatives -- explicit wave roles, half-tile stagger -- are slower:
        // f32 MFMA runs on the SIMD's FMA lanes, so a streaming wave starves its SIMD partner; see DESIGN.md.)
        auto run_sub = [&](f32x16 &acc, f32x16 &prev, int u, bool have_prev) {
            const int t = u / SUB, st = u % SUB;
            const int cur = (t - t0) & 1;
            const f32x4 *tb = tile4 + cur * G::BUF_F4 + st * (kTileCodes * RS4);
            constexpr int NG = DP / 8;
            constexpr int G1 = NG >= 2 ? 1 : NG, G2 = NG >= 4 ? 3 : NG;
            const f32x4 *ta = tb + c * RS4 + h;
            f32x4 a[NG];
            acc = (f32x16){0};
            // |c|^2 of this lane's code for the augmented MFMA: read now so its LDS latency is not paid at the tail
            const float cnv = EUCLID ? ((const float *)tb)[c * RS + DP] : 0.0f;
            mfma_prefetch<DP>(a, ta);
            mfma_range<DP, 0, G1>(acc, a, ta, xf);
            if (have_prev) tile_epilogue<METRIC, DP>(prev, u - 1, h, p.K, lb);
            __builtin_amdgcn_sched_barrier(0);
            mfma_range<DP, G1, G2>(acc, a, ta, xf);
            if (st == 0 && t + 1 < t1) stage(t + 1, cur ^ 1);
            __builtin_amdgcn_sched_barrier(0);
            mfma_range<DP, G2, NG>(acc, a, ta, xf);
            if (EUCLID) {
                const float a_aug = h ? cnv : 1.0f;  // A[code][k=0] = 1, A[code][k=1] = |c|^2
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_aug, b_aug, acc, 0, 0, 0);
            }
            if (st == SUB - 1) __syncthreads();  // next tile landed (vmcnt(0)), everybody is done reading this one
        };

        stage(t0, 0);
        __syncthreads();
        {
            f32x16 acc0, acc1;
            int u = t0 * SUB;
            const int u1 = t1 * SUB;
            bool have_prev = false;
            for (; u + 1 < u1; u += 2) {
                run_sub(acc0, acc1, u, have_prev);
                run_sub(acc1, acc0, u + 1, true);
                have_prev = true;
            }
            if (u < u1) {
                run_sub(acc0, acc1, u, have_prev);
                tile_epilogue<METRIC, DP>(acc0, u, h, p.K, lb);
            } else if (have_prev) {
                tile_epilogue<METRIC, DP>(acc1, u - 1, h, p.K, lb);
            }
        }

        float best_s;
        int best_i;
        resolve_best<METRIC>(lb, h, best_s, best_i);

        // merge the two lane halves of each row (they saw disjoint codes)
        {
            const float os = __shfl_xor(best_s, 32);
            const int oi = __shfl_xor(best_i, 32);
            const bool take = EUCLID ? (os < best_s || (os == best_s && oi < best_i))
                                     : (os > best_s || (os == best_s && oi < best_i));
            if (take) {
                best_s = os;
                best_i = oi;
            }
        }

        if (p.mode == kModeKeys) {
            if (h == 0 && row_ok)
                atomicMin(p.keys + (long long)head * p.M + row, make_key<METRIC>(best_s, p.idx_offset + best_i));
            continue;
        }

        if (h == 0 && row_ok) {
            const long long o = (long long)head * p.idx_hs + row * p.idx_rs + (long long)q * p.idx_qs;
            p.idx[o] = best_i;
            if (p.best) p.best[o] = best_s;
        }
        sidx[(wave * p.Q + q) * 32 + c] = best_i;

        // fragment-layout gather of the winner -> next-stage residual  (residual_vq.py:232)
        if (MULTI && q + 1 < p.Q) {
            const float *prow = pk + (long long)best_i * RS + 4 * h;
            const float qs = EUCLID ? -0.5f : 1.0f;  // undo the packed pre-scale (exact)
            if (p.ste) {
#pragma unroll
                for (int g = 0; g < DP / 8; ++g) {
                    const f32x4 pv = *(const f32x4 *)(prow + 8 * g);
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const float r = xf[4 * g + u];
                        const float quant = r + (qs * pv[u] - r);  // the value the layer returns in train mode
                        xf[4 * g + u] = r - quant;
                    }
                    if ((g & 3) == 3) __builtin_amdgcn_sched_barrier(0);  // bound the loads in flight (VGPRs)
                }
            } else {
#pragma unroll
                for (int g = 0; g < DP / 8; ++g) {
                    const f32x4 pv = *(const f32x4 *)(prow + 8 * g);
#pragma unroll
                    for (int u = 0; u < 4; ++u) xf[4 * g + u] = xf[4 * g + u] - qs * pv[u];
                    if ((g & 3) == 3) __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }

    STAMP(2);
    __builtin_amdgcn_s_setprio(2);
    if (p.mode == kModeKeys) return;
    if (p.out == nullptr && p.loss_part == nullptr) return;

    // ---------------- fused finalize (natural layout) ----------------
    //   quant_q = c_q[idx_q]                       (eval)      vector_quantize_pytorch.py:227
    //           = r_q + (c_q[idx_q] - r_q)         (train)     vector_quantize_pytorch.py:273
    //   r_{q+1} = r_q - quant_q ; out = ((0 + quant_1) + quant_2) + ...   residual_vq.py:232-233
    //   sq_err_q += (c_q[idx_q] - r_q)^2                                   vector_quantize_pytorch.py:362
    const bool need_r = p.ste || p.loss_part;
    float *lerr = (float *)(sidx + WAVES * p.Q * 32) + (wave * p.Q) * 64 + lane;  // [WAVES][Q][64], MULTI only
    if (MULTI && p.loss_part)
        for (int q = 0; q < p.Q; ++q) lerr[q * 64] = 0.0f;
    float e0 = 0.0f;
    const float *cbh = p.cb + (long long)head * p.cb_hs;
    float *outh = p.out ? p.out + (long long)head * p.out_hs : nullptr;
    const int nrows = (p.M - row0 >= 32) ? 32 : (int)(p.M - row0);  // wave-uniform, >= 1 ... rows of this wave
    constexpr int RB = 4;  // rows in flight: the gathers are latency-bound, so issue RB rows' loads before using them
    for (int rr0 = 0; rr0 < (p.vec_fin ? nrows : 0); rr0 += RB) {
        f32x4 r[RB][G::NCH4], o[RB][G::NCH4];
#pragma unroll
        for (int k = 0; k < RB; ++k) {
            const int rr = (rr0 + k < nrows) ? rr0 + k : nrows - 1;  // clamp: duplicates are computed, not stored
            const float *xr = xh + (row0 + rr) * p.x_rs;
#pragma unroll
            for (int j = 0; j < G::NCH4; ++j) {
                const int d = 4 * (lane + 64 * j);
                o[k][j] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
                r[k][j] = o[k][j];
                if (need_r && d < p.D) r[k][j] = *(const f32x4 *)(xr + d);
            }
        }
        for (int q = 0; q < (MULTI ? p.Q : 1); ++q) {
            f32x4 cv[RB][G::NCH4];
#pragma unroll
            for (int k = 0; k < RB; ++k) {
                const int rr = (rr0 + k < nrows) ? rr0 + k : nrows - 1;
                const int i = sidx[(wave * p.Q + q) * 32 + rr];
                const float *crow = cbh + (long long)q * p.cb_qs + (long long)i * p.D;
#pragma unroll
                for (int j = 0; j < G::NCH4; ++j) {
                    const int d = 4 * (lane + 64 * j);
                    cv[k][j] = (d < p.D) ? *(const f32x4 *)(crow + d) : (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
                }
            }
            float e = 0.0f;
#pragma unroll
            for (int k = 0; k < RB; ++k) {
                const bool live = rr0 + k < nrows;
#pragma unroll
                for (int j = 0; j < G::NCH4; ++j) {
                    f32x4 quant = cv[k][j];
                    if (need_r) {
                        const f32x4 diff = cv[k][j] - r[k][j];
                        if (live) {
                            e = fmaf(diff.x, diff.x, e);
                            e = fmaf(diff.y, diff.y, e);
                            e = fmaf(diff.z, diff.z, e);
                            e = fmaf(diff.w, diff.w, e);
                        }
                        if (p.ste) quant = r[k][j] + diff;
                        r[k][j] = r[k][j] - quant;
                    }
                    o[k][j] = o[k][j] + quant;
                }
            }
            if (MULTI) {
                if (p.loss_part) lerr[q * 64] += e;
            } else {
                e0 += e;
            }
        }
        if (outh) {
#pragma unroll
            for (int k = 0; k < RB; ++k) {
                if (rr0 + k < nrows) {
                    float *orow = outh + (row0 + rr0 + k) * p.out_rs;
#pragma unroll
                    for (int j = 0; j < G::NCH4; ++j) {
                        const int d = 4 * (lane + 64 * j);
                        if (d < p.D) __builtin_nontemporal_store(o[k][j], (f32x4 *)(orow + d));  // streamed once, never re-read
                    }
                }
            }
        }
    }
    for (int rr = 0; rr < (p.vec_fin ? 0 : nrows); ++rr) {
        const long long grow = row0 + rr;
        const float *xr = xh + grow * p.x_rs;
        float *orow = outh ? outh + grow * p.out_rs : nullptr;
        {
            float r[G::NEL], o[G::NEL];
#pragma unroll
            for (int j = 0; j < G::NEL; ++j) {
                const int d = lane + 64 * j;
                o[j] = 0.0f;
                r[j] = (need_r && d < p.D) ? xr[d] : 0.0f;
            }
            for (int q = 0; q < (MULTI ? p.Q : 1); ++q) {
                const int i = sidx[(wave * p.Q + q) * 32 + rr];
                const float *crow = cbh + (long long)q * p.cb_qs + (long long)i * p.D;
                float e = 0.0f;
#pragma unroll
                for (int j = 0; j < G::NEL; ++j) {
                    const int d = lane + 64 * j;
                    if (d < p.D) {
                        const float cv = crow[d];
                        float quant = cv;
                        if (need_r) {
                            const float diff = cv - r[j];
                            e = fmaf(diff, diff, e);
                            if (p.ste) quant = r[j] + diff;
                            r[j] = r[j] - quant;
                        }
                        o[j] = o[j] + quant;
                    }
                }
                if (MULTI) {
                    if (p.loss_part) lerr[q * 64] += e;
                } else {
                    e0 += e;
                }
            }
            if (orow) {
#pragma unroll
                for (int j = 0; j < G::NEL; ++j) {
                    const int d = lane + 64 * j;
                    if (d < p.D) orow[d] = o[j];
                }
            }
        }
    }
    STAMP(3);
    if (p.loss_part) {
        float *lp = p.loss_part + (((long long)head * gridDim.x + blockIdx.x) * WAVES + wave) * p.Q;
        for (int q = 0; q < (MULTI ? p.Q : 1); ++q) {
            float e = MULTI ? lerr[q * 64] : e0;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) e += __shfl_xor(e, o);
            if (lane == 0) lp[q] = e;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// similarity consumers: the same MFMA sweep with two other epilogues.
//   kAuxSims   write sim[h, m, k] = -sqrt(max(0, t)) (Euclid) / x.c (dot) -- the third return value of
//              Codebook.forward (codebooks.py:386,435), bit-identical to the values the search compares.
//   kAuxStats  online softmax over logits = scale * sim: per row log-sum-exp and the logit of a given target code
//              (F.cross_entropy(distances, codes) -- vector_quantize_pytorch.py:287-297 -- without [M, K] in memory).
// Performance is secondary here (training-only losses): plain loop, no in-wave software pipeline.
// ------------------------------------------------------------------------------------------------
constexpr int kAuxSims = 0;
constexpr int kAuxStats = 1;

struct AuxParams {
    const float *x;
    long long x_rs, x_hs;
    const float *packed;
    long long pk_hs;
    unsigned pk_bytes;
    long long M;
    int K, D, ntiles, vec_x;
    float *sims;  // kAuxSims
    long long sims_rs, sims_hs;
    int vec_s;
    float scale;  // kAuxStats
    const long long *target;
    long long tgt_rs, tgt_hs;
    float *lse, *tgt_logit;  // [H * M]
};

// x rows of one wave -> MFMA B fragments (same layout and |x|^2 chain as the search kernel's prologue)
template <int DP, int WAVES, bool EUCLID>
__device__ __forceinline__ void load_x_fragments(const float *xh, long long x_rs, long long M, int D, int vec_x,
                                                 long long row0, float *smem, int wave, int lane,
                                                 float (&xf)[DP / 2], float &xn0) {
    using G = Geo<DP, WAVES>;
    constexpr int CH = G::CH, XS = G::XS;
    const int c = lane & 31, h = lane >> 5;
    float *xs = smem + wave * (32 * XS);
    constexpr int NCHUNK = DP / CH;
    constexpr int LPL = CH / 8;
    xn0 = 0.0f;
    f32x4 v[2][LPL];  // global loads of chunk i+1 are in flight while chunk i goes through LDS
    auto load_chunk = [&](int ch, f32x4 (&dst)[LPL]) {
#pragma unroll
        for (int it = 0; it < LPL; ++it) {
            const int f = it * 64 + lane;
            const int r = f / (CH / 4), c4 = f % (CH / 4);
            long long grow = row0 + r;
            if (grow >= M) grow = M - 1;
            const int d0 = ch * CH + c4 * 4;
            const float *src = xh + grow * x_rs + d0;
            f32x4 t = {0.0f, 0.0f, 0.0f, 0.0f};
            if (vec_x) {
                if (d0 < D) t = *(const f32x4 *)src;
            } else {
                if (d0 + 0 < D) t.x = src[0];
                if (d0 + 1 < D) t.y = src[1];
                if (d0 + 2 < D) t.z = src[2];
                if (d0 + 3 < D) t.w = src[3];
            }
            dst[it] = t;
        }
    };
    load_chunk(0, v[0]);
#pragma unroll
    for (int ch = 0; ch < NCHUNK; ++ch) {
        if (ch + 1 < NCHUNK) load_chunk(ch + 1, v[(ch + 1) & 1]);
#pragma unroll
        for (int it = 0; it < LPL; ++it) {
            const int f = it * 64 + lane;
            const int r = f / (CH / 4), c4 = f % (CH / 4);
            *(f32x4 *)(xs + r * XS + c4 * 4) = v[ch & 1][it];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const float *rp = xs + c * XS;
#pragma unroll
        for (int j = 0; j < CH / 8; ++j) {
            const f32x4 lo = *(const f32x4 *)(rp + 8 * j);
            const f32x4 hi = *(const f32x4 *)(rp + 8 * j + 4);
            if (EUCLID) {
                xn0 = fmaf(lo.x, lo.x, xn0);
                xn0 = fmaf(lo.y, lo.y, xn0);
                xn0 = fmaf(lo.z, lo.z, xn0);
                xn0 = fmaf(lo.w, lo.w, xn0);
                xn0 = fmaf(hi.x, hi.x, xn0);
                xn0 = fmaf(hi.y, hi.y, xn0);
                xn0 = fmaf(hi.z, hi.z, xn0);
                xn0 = fmaf(hi.w, hi.w, xn0);
                asm volatile("" : "+v"(xn0));
            }
            const f32x4 m = h ? hi : lo;
            const auto xy = __builtin_amdgcn_permlane32_swap(__float_as_uint(m.x), __float_as_uint(m.y), false, false);
            const auto zw = __builtin_amdgcn_permlane32_swap(__float_as_uint(m.z), __float_as_uint(m.w), false, false);
            const int sb = ch * (CH / 2) + 4 * j;
            xf[sb + 0] = __uint_as_float(xy[0]);
            xf[sb + 1] = __uint_as_float(zw[0]);
            xf[sb + 2] = __uint_as_float(xy[1]);
            xf[sb + 3] = __uint_as_float(zw[1]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __syncthreads();
}

template <int DP, int WAVES, int METRIC, int MODE>
__global__ void __launch_bounds__(WAVES * 64, (DP <= 256 ? 2 : 1)) vq_sweep_aux(const AuxParams p) {
    using G = Geo<DP, WAVES>;
    constexpr int RS = G::RS, RS4 = G::RS4, SUB = G::SUB, NG = DP / 8;
    constexpr bool EUCLID = (METRIC == VQ_METRIC_EUCLID);

    extern __shared__ __attribute__((aligned(16))) float smem[];
    f32x4 *tile4 = (f32x4 *)smem;
    lds_f32x4 *tile4_lds = (lds_f32x4 *)smem;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 31, h = lane >> 5;
    const int head = blockIdx.y;
    const long long row0 = ((long long)blockIdx.x * WAVES + wave) * 32;
    const float *xh = p.x + (long long)head * p.x_hs;
    const float INF = __builtin_inff();

    float xf[DP / 2];
    float xn0;
    load_x_fragments<DP, WAVES, EUCLID>(xh, p.x_rs, p.M, p.D, p.vec_x, row0, smem, wave, lane, xf, xn0);

    const long long row = row0 + c;
    const bool row_ok = row < p.M;
    const float *pk = p.packed + (long long)head * p.pk_hs;
    const float b_aug = h ? 1.0f : xn0;

    // kAuxStats state: running max / sum of exp over the codes this lane has seen, and the target's logit
    float run_m = -INF, run_s = 0.0f, tgt_l = -INF;
    int tgt = -1;
    if (MODE == kAuxStats && p.target && row_ok) {
        const long long tv = p.target[(long long)head * p.tgt_hs + row * p.tgt_rs];
        tgt = (tv >= 0 && tv < p.K) ? (int)tv : (tv < 0 ? -1 : -2);  // -2: out of range -> logit stays -inf
    }
    // sub-tile and register that hold the target for THIS lane (-1: never)
    const int tgt_u = (tgt >= 0 && ((tgt >> 2) & 1) == h) ? (tgt >> 5) : -1;
    const int tgt_r = (tgt & 3) + 4 * ((tgt >> 3) & 3);

    auto stage = [&](int tile, int buf) {
#pragma unroll
        for (int i = 0; i < (G::TILE_CHUNKS + WAVES - 1) / WAVES; ++i) {
            const int ck = i * WAVES + wave;
            if (ck < G::TILE_CHUNKS)
                lds_dma16(pk, p.pk_bytes, lane * 16, (tile * G::TILE_F4 + ck * 64) * 16,
                          tile4_lds + buf * G::BUF_F4 + ck * 64);
        }
    };

    stage(0, 0);
    __syncthreads();
    for (int t = 0; t < p.ntiles; ++t) {
        const int cur = t & 1;
        if (t + 1 < p.ntiles) stage(t + 1, cur ^ 1);
#pragma unroll 1
        for (int st = 0; st < SUB; ++st) {
            const int u = t * SUB + st;
            if (u * kTileCodes >= p.K) break;  // wave-uniform: nothing but padding from here on
            const f32x4 *tb = tile4 + cur * G::BUF_F4 + st * (kTileCodes * RS4);
            const f32x4 *ta = tb + c * RS4 + h;
            f32x4 a[NG];
            f32x16 acc = {0};
            const float cnv = EUCLID ? ((const float *)tb)[c * RS + DP] : 0.0f;
            mfma_prefetch<DP>(a, ta);
            mfma_range<DP, 0, NG>(acc, a, ta, xf);
            if (EUCLID) {
                const float a_aug = h ? cnv : 1.0f;
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_aug, b_aug, acc, 0, 0, 0);
            }
            const int cbase = u * kTileCodes + 4 * h;
            if (MODE == kAuxSims) {
                float *srow = p.sims + (long long)head * p.sims_hs + row * p.sims_rs;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float tv = acc[4 * g + e];
                        v[e] = EUCLID ? -sqrtf(fmaxf(tv, 0.0f)) : tv;  // correctly rounded, like the search
                    }
                    const int code = cbase + 8 * g;
                    if (row_ok) {
                        if (p.vec_s) {
                            if (code < p.K) *(f32x4 *)(srow + code) = v;
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (code + e < p.K) srow[code + e] = v[e];
                        }
                    }
                }
            } else {
                float l[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float tv = acc[r];
                    l[r] = EUCLID ? -p.scale * __builtin_amdgcn_sqrtf(fmaxf(tv, 0.0f)) : p.scale * tv;
                }
                if (u * kTileCodes + kTileCodes > p.K) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (cbase + (r & 3) + 8 * (r >> 2) >= p.K) l[r] = -INF;
                }
                if (u == tgt_u) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) tgt_l = (r == tgt_r) ? l[r] : tgt_l;
                }
                float tm = l[0];
#pragma unroll
                for (int r = 1; r < 16; ++r) tm = fmaxf(tm, l[r]);
                if (tm > run_m) {
                    run_s *= __expf(run_m - tm);  // exp(-inf) = 0 on the first visit
                    run_m = tm;
                }
                if (run_m > -INF) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) run_s += __expf(l[r] - run_m);
                }
            }
        }
        __syncthreads();  // next tile landed, everybody is done reading this one
    }

    if (MODE == kAuxStats) {
        const float om = __shfl_xor(run_m, 32), os = __shfl_xor(run_s, 32), ot = __shfl_xor(tgt_l, 32);
        const float mm = fmaxf(run_m, om);  // lane half 0 always saw code 0, so mm is finite
        const float s = (run_m > -INF ? run_s * __expf(run_m - mm) : 0.0f) + (om > -INF ? os * __expf(om - mm) : 0.0f);
        if (h == 0 && row_ok) {
            const long long o = (long long)head * p.M + row;
            p.lse[o] = mm + __logf(s);
            if (p.tgt_logit) p.tgt_logit[o] = (tgt == -1) ? 0.0f : fmaxf(tgt_l, ot);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// fused cross-entropy backward:  d/dx of  coef * sum_rows (logsumexp_k sim - sim[target])
//   S sweep      t[code, row] exactly as in the search (codes on the MFMA i axis, rows on j)
//   epilogue     p = exp(sim - lse[row]);  gs = coef * (p - [code == target])            (d loss / d sim)
//                Euclid: ratio = gs / sim (0 where sim == 0)   -- ATen _euclidean_dist_backward with dist = -sim
//   G sweep      the 16 accumulator registers of the S sweep are, as they stand, valid B operands of
//                v_mfma_f32_32x32x2_f32 for the second contraction (register r of the two lane halves = one k-pair of
//                codes), so  G[pos, row] += Cimg[code, pos] * ratio[code, row]  needs no shuffles: Dp/2 more MFMAs per
//                sub-tile, A fragments read from the SAME LDS tile (one ds_read_b128 feeds 4 MFMAs, "virtual" d-chunks
//                of stride 4).
//   finalize     Euclid: gx = x * sum_k ratio + 0.5 * G   (the image holds -2c);   dot: gx = G.   G is staged through
//                LDS to undo the fragment / even-odd layout and written with coalesced stores.
// Register budget: Dp/2 (x fragments) + Dp/2 (G accumulators) + ~60 -> 4-wave workgroups; Dp = 256 runs one wave per
// SIMD (512 registers), Dp <= 128 two.  Dp = 512: two workgroups per row block (blockIdx.z), each repeats the S sweep
// and produces one 256-wide half of the dims (3 instead of 2 units of MFMA work).
// ------------------------------------------------------------------------------------------------
struct CeBwdParams {
    const float *x;
    long long x_rs, x_hs;
    const float *packed;
    long long pk_hs;
    unsigned pk_bytes;
    long long M;
    int K, D, ntiles, vec_x;
    const float *lse;  // [H * M]
    const long long *target;
    long long tgt_rs, tgt_hs;
    const float *coef;  // one float on the device: upstream gradient / number of non-ignored rows
    const float *cb;    // natural codebook (the target's term is added in the finalize)
    long long cb_hs;
    const float *tgt_logit;  // [H * M] similarity of the target code (vq_softmax_stats_f32 output, scale 1)
    float *gx;
    long long gx_rs, gx_hs;
};

template <int DP>
struct CeGeo {
    static constexpr int V = DP >= 128 ? 4 : DP / 32;  // floats per A-fragment read (positions 4i+e / 2i+e / i)
    static constexpr int NH = DP > 256 ? DP / 256 : 1; // the G accumulators of Dp = 512 do not fit beside the x fragments:
                                                       // blockIdx.z picks a 256-wide half of the packed positions
    static constexpr int WID = DP / NH;                // positions (= dims) one workgroup produces
    static constexpr int NJ = WID / (32 * V);          // 128-wide (V = 4) position blocks
    static constexpr int NACC = WID / 32;              // 32x32 accumulators of G
    static constexpr int GS = WID + 4;                 // staging row stride (floats): 16-B aligned rows, 2-way conflicts on the column writes
};

template <int DP, int METRIC>
__global__ void __launch_bounds__(256, (DP <= 128 ? 2 : 1)) vq_ce_backward(const CeBwdParams p) {
    constexpr int WAVES = 4;
    using G = Geo<DP, WAVES>;
    using CG = CeGeo<DP>;
    constexpr int RS = G::RS, RS4 = G::RS4, SUB = G::SUB, NG = DP / 8, V = CG::V, NJ = CG::NJ, NACC = CG::NACC;
    constexpr int WID = CG::WID;
    constexpr bool EUCLID = (METRIC == VQ_METRIC_EUCLID);
    const int pos0 = (CG::NH > 1) ? (int)blockIdx.z * WID : 0;  // first packed position (= dim) of this workgroup
    constexpr float LOG2E = 1.4426950408889634f;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    f32x4 *tile4 = (f32x4 *)smem;
    lds_f32x4 *tile4_lds = (lds_f32x4 *)smem;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 31, h = lane >> 5;
    const int head = blockIdx.y;
    const long long row0 = ((long long)blockIdx.x * WAVES + wave) * 32;
    const float *xh = p.x + (long long)head * p.x_hs;

    STAMP(0);
    float xf[DP / 2];
    float xn0;
    load_x_fragments<DP, WAVES, EUCLID>(xh, p.x_rs, p.M, p.D, p.vec_x, row0, smem, wave, lane, xf, xn0);
    STAMP(1);

    const long long row = row0 + c;
    const bool row_ok = row < p.M;
    const float *pk = p.packed + (long long)head * p.pk_hs;
    const float b_aug = h ? 1.0f : xn0;

    int tgt = -1;
    float lse2 = 0.0f;  // lse * log2(e)
    float tlog = 0.0f;  // similarity of the target code
    if (row_ok) {
        const long long tv = p.target[(long long)head * p.tgt_hs + row * p.tgt_rs];
        tgt = (tv >= 0 && tv < p.K) ? (int)tv : -1;
        lse2 = p.lse[(long long)head * p.M + row] * LOG2E;
        tlog = p.tgt_logit[(long long)head * p.M + row];
    }
    const float coef_row = (tgt >= 0) ? p.coef[0] : 0.0f;  // ignored / padding rows contribute nothing
    // The sweep below handles the softmax part  coef * p_k  of d loss / d sim_k for every code alike; the one-hot part
    // (-coef at k = target) is a rank-one term per row and is added in the finalize from the natural codebook.

    f32x16 gacc[NACC];
#pragma unroll
    for (int j = 0; j < NACC; ++j) gacc[j] = (f32x16){0};
    float sum_ratio = 0.0f;

    auto stage = [&](int tile, int buf) {
#pragma unroll
        for (int i = 0; i < (G::TILE_CHUNKS + WAVES - 1) / WAVES; ++i) {
            const int ck = i * WAVES + wave;
            if (ck < G::TILE_CHUNKS)
                lds_dma16(pk, p.pk_bytes, lane * 16, (tile * G::TILE_F4 + ck * 64) * 16,
                          tile4_lds + buf * G::BUF_F4 + ck * 64);
        }
    };

    stage(0, 0);
    __syncthreads();
    for (int t = 0; t < p.ntiles; ++t) {
        const int cur = t & 1;
        if (t + 1 < p.ntiles) stage(t + 1, cur ^ 1);
#pragma unroll 1
        for (int st = 0; st < SUB; ++st) {
            const int u = t * SUB + st;
            if (u * kTileCodes >= p.K) break;  // workgroup-uniform
            const f32x4 *tb = tile4 + cur * G::BUF_F4 + st * (kTileCodes * RS4);
            // ---- S sweep
            f32x16 acc = {0};
            {
                const f32x4 *ta = tb + c * RS4 + h;
                f32x4 a[NG];
                const float cnv = EUCLID ? ((const float *)tb)[c * RS + DP] : 0.0f;
                mfma_prefetch<DP>(a, ta);
                mfma_range<DP, 0, NG>(acc, a, ta, xf);
                if (EUCLID) {
                    const float a_aug = h ? cnv : 1.0f;
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_aug, b_aug, acc, 0, 0, 0);
                }
            }
            // ---- epilogue: acc[r] <- ratio (Euclid) / d loss / d sim (dot) of code u*32 + 4h + (r&3) + 8(r>>2)
            // (softmax part only; two values per step so that the multiplies / fma / add pack into v_pk_* instructions)
            const int cbase = u * kTileCodes + 4 * h;
            const bool tail = u * kTileCodes + kTileCodes > p.K;
            bool degenerate = false;  // some squared distance <= 0: ATen gives those codes the subgradient 0
            if (EUCLID) {
                float tm = vmin3(acc[0], acc[1], acc[2]);
                tm = vmin3(tm, acc[3], acc[4]);
                tm = vmin3(tm, acc[5], acc[6]);
                tm = vmin3(tm, acc[7], acc[8]);
                tm = vmin3(tm, acc[9], acc[10]);
                tm = vmin3(tm, acc[11], acc[12]);
                tm = vmin3(tm, acc[13], acc[14]);
                tm = fminf(tm, acc[15]);
                degenerate = __any(tm <= 0.0f);
            }
            {
                const f32x2 ncoef = {-coef_row, -coef_row}, pcoef = {coef_row, coef_row};
                const f32x2 nl2e = {-LOG2E, -LOG2E}, pl2e = {LOG2E, LOG2E}, nlse = {-lse2, -lse2};
                f32x2 sum2 = {0.0f, 0.0f};
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const f32x2 tv = {acc[r], acc[r + 1]};
                    f32x2 v;
                    if (EUCLID) {
                        const f32x2 tc = {vmax3(tv.x, 0.0f, 0.0f), vmax3(tv.y, 0.0f, 0.0f)};
                        const f32x2 rs = {__builtin_amdgcn_rsqf(tc.x), __builtin_amdgcn_rsqf(tc.y)};  // 1 / dist
                        const f32x2 ex = __builtin_elementwise_fma(tc * rs, nl2e, nlse);               // (-dist - lse) log2 e
                        const f32x2 pr = {__builtin_amdgcn_exp2f(ex.x), __builtin_amdgcn_exp2f(ex.y)};
                        v = (pr * rs) * ncoef;  // gs / sim with sim = -dist; inf / nan where dist == 0 (fixed below)
                    } else {
                        const f32x2 ex = __builtin_elementwise_fma(tv, pl2e, nlse);
                        const f32x2 pr = {__builtin_amdgcn_exp2f(ex.x), __builtin_amdgcn_exp2f(ex.y)};
                        v = pr * pcoef;
                    }
                    acc[r] = v.x;
                    acc[r + 1] = v.y;
                    if (EUCLID) sum2 += v;
                }
                float sum_u = sum2.x + sum2.y;
                // rare wave-uniform fix-ups, kept out of the straight-line path: zero distances and the codebook's tail
                if (degenerate || tail) {
                    asm volatile("" ::: "memory");  // keep this a branch (16 selects per lane otherwise)
                    sum_u = 0.0f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float v = acc[r];
                        if (!(__builtin_fabsf(v) < __builtin_inff())) v = 0.0f;            // 1 / 0: subgradient 0
                        if (cbase + (r & 3) + 8 * (r >> 2) >= p.K) v = 0.0f;               // padding codes
                        acc[r] = v;
                        sum_u += v;
                    }
                }
                if (EUCLID) sum_ratio += sum_u;
            }
            // ---- G sweep: gacc[J*V + e][pos-in-chunk i, row] += Cimg[code(r, half)][128J + 4i + e] * acc[r]
            const float *trow = (const float *)tb + (4 * h) * RS + V * c + pos0;
            if constexpr (V == 4) {
                // fragment reads run PF steps ahead of their MFMAs, order pinned (same scheme as mfma_range)
                constexpr int NSEQ = NJ * 16, PF = 4;
                auto frag = [&](int n) -> f32x4 {
                    const int J = n >> 4, r = n & 15;
                    return *(const f32x4 *)(trow + ((r & 3) + 8 * (r >> 2)) * RS + 128 * J);
                };
                f32x4 af[NSEQ];
#pragma unroll
                for (int n = 0; n < PF; ++n) af[n] = frag(n);
#pragma unroll
                for (int n = 0; n < NSEQ; ++n) {
                    if (n + PF < NSEQ) af[n + PF] = frag(n + PF);
                    const int J = n >> 4, r = n & 15;
                    gacc[J * 4 + 0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[n].x, acc[r], gacc[J * 4 + 0], 0, 0, 0);
                    gacc[J * 4 + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[n].y, acc[r], gacc[J * 4 + 1], 0, 0, 0);
                    gacc[J * 4 + 2] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[n].z, acc[r], gacc[J * 4 + 2], 0, 0, 0);
                    gacc[J * 4 + 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[n].w, acc[r], gacc[J * 4 + 3], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float *ap = trow + ((r & 3) + 8 * (r >> 2)) * RS;
                    if (V == 2) {
                        const float a0 = ap[0], a1 = ap[1];
                        gacc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, acc[r], gacc[0], 0, 0, 0);
                        gacc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, acc[r], gacc[1], 0, 0, 0);
                    } else {
                        gacc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[0], acc[r], gacc[0], 0, 0, 0);
                    }
                    if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);  // bound the fragment reads in flight
                }
            }
        }
        __syncthreads();  // next tile landed, everybody is done reading this one
    }

    STAMP(2);
    // ---------------- finalize: fragment layout -> natural rows through LDS ----------------
    // (the loop's last barrier guarantees nobody reads the tile buffers any more)
    constexpr int GS = CG::GS;
    float *stg = smem + wave * (32 * GS + 96);
    float *srs = stg + 32 * GS;
    // per row: sum of the ratios (Euclid) and the one-hot term's factor  f:  gx += f * (c_target - x)  (Euclid,
    // f = -coef / dist_t = coef / sim_t, 0 at dist_t == 0)   or   gx += f * c_target  (dot, f = -coef)
    int *stg_t = (int *)(srs + 32);
    float *stg_f = srs + 64;
    if (EUCLID) sum_ratio += __shfl_xor(sum_ratio, 32);
    if (h == 0) {
        srs[c] = sum_ratio;
        stg_t[c] = tgt;
        stg_f[c] = EUCLID ? ((tlog < 0.0f) ? coef_row / tlog : 0.0f) : -coef_row;
    }
#pragma unroll
    for (int a = 0; a < NACC; ++a) {
        const int J = a / V, e = a % V;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = 8 * (r >> 2) + 4 * h + (r & 3);       // MFMA i index held by this lane's register r
            const int pos = 32 * V * J + V * i + e;             // position in this workgroup's part of the packed row
            const int p8 = pos & 7;
            const int dim = (pos & ~7) + (p8 < 4 ? 2 * p8 : 2 * (p8 - 4) + 1);
            stg[c * GS + dim] = gacc[a][r];
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // wave-private region: in-order LDS, no barrier needed
    const int nrows = (p.M - row0 >= 32) ? 32 : (int)(p.M - row0);
    float *gxh = p.gx + (long long)head * p.gx_hs;
    const float *cbh = p.cb + (long long)head * p.cb_hs;
    const bool vec_g = p.vec_x && (p.gx_rs % 4 == 0) && (p.gx_hs % 4 == 0) && (((uintptr_t)p.gx & 15) == 0) &&
                       (p.cb_hs % 4 == 0) && (((uintptr_t)p.cb & 15) == 0);
    if (vec_g) {
        constexpr int RB = 8;                      // rows in flight: the x / codebook loads are latency-bound
        constexpr int NV = (WID + 255) / 256;      // float4 per lane and row
        for (int rr0 = 0; rr0 < nrows; rr0 += RB) {
            f32x4 xv[RB][NV], cv[RB][NV];
#pragma unroll
            for (int k = 0; k < RB; ++k) {
                const int rr = (rr0 + k < nrows) ? rr0 + k : nrows - 1;
                const float *xr = xh + (row0 + rr) * p.x_rs + pos0;
                const int t = stg_t[rr];
                const float *cr = cbh + (long long)(t < 0 ? 0 : t) * p.D + pos0;
#pragma unroll
                for (int j = 0; j < NV; ++j) {
                    const int dl = 4 * (lane + 64 * j);
                    const bool in = dl < WID && pos0 + dl < p.D;
                    xv[k][j] = in ? *(const f32x4 *)(xr + dl) : (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
                    cv[k][j] = in ? *(const f32x4 *)(cr + dl) : (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
                }
            }
#pragma unroll
            for (int k = 0; k < RB; ++k) {
                if (rr0 + k < nrows) {
                    const float sr = srs[rr0 + k], f = stg_f[rr0 + k];
                    float *gr = gxh + (row0 + rr0 + k) * p.gx_rs + pos0;
#pragma unroll
                    for (int j = 0; j < NV; ++j) {
                        const int dl = 4 * (lane + 64 * j);
                        if (dl < WID && pos0 + dl < p.D) {
                            const f32x4 gv = *(const f32x4 *)(stg + (rr0 + k) * GS + dl);
                            const f32x4 x4 = xv[k][j], c4 = cv[k][j];
                            f32x4 o;
                            if (EUCLID) {
                                o.x = fmaf(f, c4.x - x4.x, fmaf(x4.x, sr, 0.5f * gv.x));
                                o.y = fmaf(f, c4.y - x4.y, fmaf(x4.y, sr, 0.5f * gv.y));
                                o.z = fmaf(f, c4.z - x4.z, fmaf(x4.z, sr, 0.5f * gv.z));
                                o.w = fmaf(f, c4.w - x4.w, fmaf(x4.w, sr, 0.5f * gv.w));
                            } else {
                                o.x = fmaf(f, c4.x, gv.x);
                                o.y = fmaf(f, c4.y, gv.y);
                                o.z = fmaf(f, c4.z, gv.z);
                                o.w = fmaf(f, c4.w, gv.w);
                            }
                            *(f32x4 *)(gr + dl) = o;
                        }
                    }
                }
            }
        }
    } else {
        for (int rr = 0; rr < nrows; ++rr) {
            const float sr = srs[rr], f = stg_f[rr];
            const int t = stg_t[rr];
            const float *xr = xh + (row0 + rr) * p.x_rs;
            const float *cr = cbh + (long long)(t < 0 ? 0 : t) * p.D;
            float *gr = gxh + (row0 + rr) * p.gx_rs;
            for (int dl = lane; dl < WID && pos0 + dl < p.D; dl += 64) {
                const int d = pos0 + dl;
                const float gv = stg[rr * GS + dl];
                gr[d] = EUCLID ? fmaf(f, cr[d] - xr[d], fmaf(xr[d], sr, 0.5f * gv)) : fmaf(f, cr[d], gv);
            }
        }
    }
    STAMP(3);
}

// scalar fallback for the similarity matrix (D > 512, cross-check): one thread per (row, code)
template <int METRIC>
__global__ void __launch_bounds__(256) vq_sims_simple(const float *__restrict__ x, long long x_rs, long long x_hs,
                                                      const float *__restrict__ cb, long long cb_hs, long long M, int K,
                                                      int D, float *__restrict__ sims, long long sims_rs, long long sims_hs) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= M * K) return;
    const long long row = gid / K;
    const int k = (int)(gid % K);
    const int head = blockIdx.y;
    const float *xr = x + (long long)head * x_hs + row * x_rs;
    const float *cr = cb + (long long)head * cb_hs + (long long)k * D;
    float acc = 0.0f;
    if (METRIC == VQ_METRIC_EUCLID) {
        float xn = 0.0f, cn = 0.0f;
        for (int d = 0; d < D; ++d) {
            xn = fmaf(xr[d], xr[d], xn);
            cn = fmaf(cr[d], cr[d], cn);
            acc = fmaf(xr[d], -2.0f * cr[d], acc);
        }
        acc = fmaf(1.0f, xn, acc);
        acc = fmaf(cn, 1.0f, acc);
        acc = -sqrtf(fmaxf(acc, 0.0f));
    } else {
        for (int d = 0; d < D; ++d) acc = fmaf(xr[d], cr[d], acc);
    }
    sims[(long long)head * sims_hs + row * sims_rs + k] = acc;
}

// ------------------------------------------------------------------------------------------------
// scalar-FMA fallback search: one thread per row, any D / K.  Same chain order as the MFMA kernel
// (and the oracle), so it doubles as an on-device cross-check.  Emits packed keys.
// ------------------------------------------------------------------------------------------------
template <int METRIC>
__global__ void vq_search_simple(const float *__restrict__ x, long long x_rs, long long x_hs,
                                 const float *__restrict__ cb, long long cb_hs, long long M, int K, int D,
                                 long long idx_offset, long long *__restrict__ keys) {
    const long long row = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= M) return;
    const int head = blockIdx.y;
    const float *xr = x + (long long)head * x_hs + row * x_rs;
    const float *c0 = cb + (long long)head * cb_hs;
    float xn = 0.0f;
    if (METRIC == VQ_METRIC_EUCLID)
        for (int d = 0; d < D; ++d) xn = fmaf(xr[d], xr[d], xn);
    float best = (METRIC == VQ_METRIC_EUCLID) ? __builtin_inff() : -__builtin_inff();
    int bi = 0;
    for (int k = 0; k < K; ++k) {
        const float *cr = c0 + (long long)k * D;
        float acc = 0.0f;
        if (METRIC == VQ_METRIC_EUCLID) {
            float cn = 0.0f;
            for (int d = 0; d < D; ++d) {
                const float cv = cr[d];
                acc = fmaf(xr[d], -2.0f * cv, acc);
                cn = fmaf(cv, cv, cn);
            }
            acc = fmaf(1.0f, xn, acc);
            acc = fmaf(cn, 1.0f, acc);
            const float s = sqrtf(fmaxf(acc, 0.0f));
            if (s < best) {
                best = s;
                bi = k;
            }
        } else {
            for (int d = 0; d < D; ++d) acc = fmaf(xr[d], cr[d], acc);
            if (acc > best) {
                best = acc;
                bi = k;
            }
        }
    }
    atomicMin(keys + (long long)head * M + row, make_key<METRIC>(best, idx_offset + bi));
}

__global__ void vq_keys_init_kernel(long long *keys, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) keys[i] = 0x7FFFFFFFFFFFFFFFll;
}

// ------------------------------------------------------------------------------------------------
// finalize from keys: one wave per row (grid-strided), natural layout
// ------------------------------------------------------------------------------------------------
struct FinalizeParams {
    const long long *keys;
    const float *x;
    long long x_rs, x_hs;
    const float *cb;
    long long cb_hs;
    float *out;
    long long out_rs, out_hs;
    long long *idx;
    long long idx_rs, idx_hs;
    float *best;
    float *loss_part;  // [H][gridDim.x * 4]
    long long M;
    int D, metric, ste;
};

__global__ void __launch_bounds__(256) vq_finalize_kernel(const FinalizeParams p) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int head = blockIdx.y;
    const long long nw = (long long)gridDim.x * 4;
    float e = 0.0f;
    for (long long row = (long long)blockIdx.x * 4 + wave; row < p.M; row += nw) {
        const long long key = p.keys[(long long)head * p.M + row];
        const long long i = (long long)(key & 0xFFFFFFFFll);
        if (lane == 0) {
            const long long o = (long long)head * p.idx_hs + row * p.idx_rs;
            if (p.idx) p.idx[o] = i;
            if (p.best) p.best[o] = key_value(key, p.metric);
        }
        const bool need_x = p.ste || p.loss_part;
        if (!p.out && !p.loss_part) continue;
        const float *xr = p.x + (long long)head * p.x_hs + row * p.x_rs;
        const float *crow = p.cb + (long long)head * p.cb_hs + i * p.D;
        float *orow = p.out ? p.out + (long long)head * p.out_hs + row * p.out_rs : nullptr;
        for (int d = lane; d < p.D; d += 64) {
            const float cv = crow[d];
            float quant = cv;
            if (need_x) {
                const float r = xr[d];
                const float diff = cv - r;
                e = fmaf(diff, diff, e);
                if (p.ste) quant = r + diff;
            }
            if (orow) orow[d] = quant;
        }
    }
    if (p.loss_part) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) e += __shfl_xor(e, o);
        if (lane == 0) p.loss_part[((long long)head * gridDim.x + blockIdx.x) * 4 + wave] = e;
    }
}

// sq_err[q] = sum over parts of loss_part[part*Q + q]   (double, fixed order)
__global__ void __launch_bounds__(256) vq_loss_reduce_kernel(const float *__restrict__ part, long long nparts, int Q,
                                                             double *__restrict__ sq_err) {
    __shared__ double sh[256];
    const int q = blockIdx.x;
    double s = 0.0;
    for (long long i = threadIdx.x; i < nparts; i += 256) s += (double)part[i * Q + q];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) sq_err[q] = sh[0];
}

// ------------------------------------------------------------------------------------------------
// training-state step that follows the hot path: exponential-moving-average update
//   counts[h,k]  = #rows of head h assigned to code k            (reference: embed_onehot.sum(1), codebooks.py:408)
//   sums[h,k,:]  = sum of those rows                              (einsum("h n d, h n c -> h c d"), codebooks.py:413)
// The reference builds both through the [h, M, K] one-hot tensor; here they are a scatter-add with float atomics
// shaped for the memory-side atomic units: one dword per lane, 256 contiguous bytes per wave instruction.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) vq_ema_accumulate_kernel(const float *__restrict__ x, long long x_rs, long long x_hs,
                                                                const long long *__restrict__ idx, long long idx_rs,
                                                                long long idx_hs, const unsigned char *__restrict__ mask,
                                                                long long M, int K, int D, float *__restrict__ counts,
                                                                float *__restrict__ sums) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int head = blockIdx.y;
    const long long nw = (long long)gridDim.x * 4;
    const float *xh = x + (long long)head * x_hs;
    const long long *ih = idx + (long long)head * idx_hs;
    float *ch = counts + (long long)head * K;
    float *sh = sums + (long long)head * K * D;
    for (long long row = (long long)blockIdx.x * 4 + wave; row < M; row += nw) {
        if (mask && !mask[(long long)head * M + row]) continue;  // wave-uniform
        const long long k = ih[row * idx_rs];
        if (k < 0 || k >= K) continue;
        const float *xr = xh + row * x_rs;
        float *dst = sh + k * D;
        for (int d = lane; d < D; d += 64) atomicAdd(dst + d, xr[d]);
        if (lane == 0) atomicAdd(ch + k, 1.0f);
    }
}

// cluster_size <- lerp(cluster_size, counts, 1 - decay);  total[h] = sum_k cluster_size      (codebooks.py:411,419-421)
__global__ void __launch_bounds__(256) vq_ema_sizes_kernel(float *__restrict__ cluster_size, const float *__restrict__ counts,
                                                           int K, float weight, float *__restrict__ total) {
    __shared__ float sh[256];
    const int head = blockIdx.x;
    float s = 0.0f;
    for (int k = threadIdx.x; k < K; k += 256) {
        const float old = cluster_size[(long long)head * K + k];
        const float nw = old + weight * (counts[(long long)head * K + k] - old);
        cluster_size[(long long)head * K + k] = nw;
        s += nw;
    }
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) total[head] = sh[0];
}

// embed_avg <- lerp(embed_avg, sums, 1 - decay); embeddings <- [l2norm](embed_avg / laplace-smoothed size)
// one wave per code row                                                             (codebooks.py:417-425)
__global__ void __launch_bounds__(256) vq_ema_codes_kernel(const float *__restrict__ cluster_size, const float *__restrict__ total,
                                                           float *__restrict__ embed_avg, const float *__restrict__ sums,
                                                           float *__restrict__ embeddings, int H, int K, int D, float weight,
                                                           float eps, int l2norm) {
    const int lane = threadIdx.x & 63;
    const long long rowid = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (rowid >= (long long)H * K) return;
    const int head = (int)(rowid / K);
    const float tot = total[head];
    const float cs = cluster_size[rowid];
    const float smoothed = (cs + eps) / (tot + (float)K * eps) * tot;
    float *avg = embed_avg + rowid * D;
    const float *sm = sums + rowid * D;
    float *emb = embeddings + rowid * D;
    float nrm = 0.0f;
    for (int d = lane; d < D; d += 64) {
        const float old = avg[d];
        const float a = old + weight * (sm[d] - old);
        avg[d] = a;
        const float e = a / smoothed;
        emb[d] = e;
        nrm = fmaf(e, e, nrm);
    }
    if (l2norm) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) nrm += __shfl_xor(nrm, o);
        const float inv = 1.0f / fmaxf(sqrtf(nrm), 1e-12f);
        for (int d = lane; d < D; d += 64) emb[d] = emb[d] * inv;  // same lane wrote it
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
struct DevInfo {
    int cus = 0;
    bool ok = false;
    char name[128] = "";
};

const DevInfo &dev_info() {
    static thread_local DevInfo info;
    static thread_local int cached_dev = -1;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) {
        info.ok = false;
        return info;
    }
    if (dev != cached_dev) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) == hipSuccess) {
            info.cus = prop.multiProcessorCount;
            snprintf(info.name, sizeof(info.name), "%s", prop.gcnArchName);
            info.ok = true;
            cached_dev = dev;
        } else {
            info.ok = false;
        }
    }
    return info;
}

template <int DP, int WAVES, int METRIC, bool MULTI>
int launch_search_t(const SearchParams &p, int H, int splits, hipStream_t s) {
    using G = Geo<DP, WAVES>;
    const size_t lds = (size_t)G::MAIN_FLOATS * 4 + (size_t)WAVES * p.Q * 32 * 4 +
                       ((MULTI && p.loss_part) ? (size_t)WAVES * p.Q * 64 * 4 : 0);
    if (lds > 160 * 1024) return fail(VQ_E_UNSUPPORTED, "vq_search: LDS budget exceeded (too many residual stages)");
    auto kern = vq_search_mfma<DP, WAVES, METRIC, MULTI>;
    static thread_local bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute");
        attr_done = true;
    }
    const long long rows_per_wg = 32ll * WAVES;
    dim3 grid((unsigned)((p.M + rows_per_wg - 1) / rows_per_wg), (unsigned)H, (unsigned)splits);
    hipLaunchKernelGGL(kern, grid, dim3(WAVES * 64), lds, s, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "vq_search_mfma launch");
    return 0;
}

template <int DP, int WAVES>
int launch_search_m(const SearchParams &p, int H, int splits, int metric, hipStream_t s) {
    if (p.Q > 1) {
        if (metric == VQ_METRIC_EUCLID) return launch_search_t<DP, WAVES, VQ_METRIC_EUCLID, true>(p, H, splits, s);
        return launch_search_t<DP, WAVES, VQ_METRIC_DOT, true>(p, H, splits, s);
    }
    if (metric == VQ_METRIC_EUCLID) return launch_search_t<DP, WAVES, VQ_METRIC_EUCLID, false>(p, H, splits, s);
    return launch_search_t<DP, WAVES, VQ_METRIC_DOT, false>(p, H, splits, s);
}

int launch_search(int DP, int waves, const SearchParams &p, int H, int splits, int metric, hipStream_t s) {
    switch (DP) {
        case 32: return waves == 8 ? launch_search_m<32, 8>(p, H, splits, metric, s) : launch_search_m<32, 4>(p, H, splits, metric, s);
        case 64: return waves == 8 ? launch_search_m<64, 8>(p, H, splits, metric, s) : launch_search_m<64, 4>(p, H, splits, metric, s);
        case 128: return waves == 8 ? launch_search_m<128, 8>(p, H, splits, metric, s) : launch_search_m<128, 4>(p, H, splits, metric, s);
        case 256: return waves == 8 ? launch_search_m<256, 8>(p, H, splits, metric, s) : launch_search_m<256, 4>(p, H, splits, metric, s);
        case 512: return launch_search_m<512, 4>(p, H, splits, metric, s);
    }
    return fail(VQ_E_UNSUPPORTED, "vq_search: unsupported padded dim");
}


template <int DP, int WAVES, int METRIC, int MODE>
int launch_aux_t(const AuxParams &p, int H, hipStream_t s) {
    using G = Geo<DP, WAVES>;
    const size_t lds = (size_t)G::MAIN_FLOATS * 4;
    auto kern = vq_sweep_aux<DP, WAVES, METRIC, MODE>;
    static thread_local bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute");
        attr_done = true;
    }
    const long long rows_per_wg = 32ll * WAVES;
    dim3 grid((unsigned)((p.M + rows_per_wg - 1) / rows_per_wg), (unsigned)H, 1);
    hipLaunchKernelGGL(kern, grid, dim3(WAVES * 64), lds, s, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "vq_sweep_aux launch");
    return 0;
}

template <int DP, int WAVES>
int launch_aux_m(const AuxParams &p, int H, int metric, int mode, hipStream_t s) {
    if (mode == kAuxSims) {
        if (metric == VQ_METRIC_EUCLID) return launch_aux_t<DP, WAVES, VQ_METRIC_EUCLID, kAuxSims>(p, H, s);
        return launch_aux_t<DP, WAVES, VQ_METRIC_DOT, kAuxSims>(p, H, s);
    }
    if (metric == VQ_METRIC_EUCLID) return launch_aux_t<DP, WAVES, VQ_METRIC_EUCLID, kAuxStats>(p, H, s);
    return launch_aux_t<DP, WAVES, VQ_METRIC_DOT, kAuxStats>(p, H, s);
}

int launch_aux(int DP, const AuxParams &p, int H, int metric, int mode, hipStream_t s) {
    switch (DP) {
        case 32: return launch_aux_m<32, 4>(p, H, metric, mode, s);
        case 64: return launch_aux_m<64, 4>(p, H, metric, mode, s);
        case 128: return launch_aux_m<128, 4>(p, H, metric, mode, s);
        case 256: return launch_aux_m<256, 4>(p, H, metric, mode, s);
        case 512: return launch_aux_m<512, 4>(p, H, metric, mode, s);
    }
    return fail(VQ_E_UNSUPPORTED, "vq_sweep_aux: unsupported padded dim");
}


template <int DP, int METRIC>
int launch_ce_bwd_t(const CeBwdParams &p, int H, hipStream_t s) {
    using G = Geo<DP, 4>;
    const size_t stage_floats = (size_t)4 * (32 * CeGeo<DP>::GS + 96);
    const size_t lds = 4 * ((size_t)G::MAIN_FLOATS > stage_floats ? (size_t)G::MAIN_FLOATS : stage_floats);
    auto kern = vq_ce_backward<DP, METRIC>;
    static thread_local bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute");
        attr_done = true;
    }
    dim3 grid((unsigned)((p.M + 127) / 128), (unsigned)H, (unsigned)CeGeo<DP>::NH);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "vq_ce_backward launch");
    return 0;
}

template <int DP>
int launch_ce_bwd_m(const CeBwdParams &p, int H, int metric, hipStream_t s) {
    if (metric == VQ_METRIC_EUCLID) return launch_ce_bwd_t<DP, VQ_METRIC_EUCLID>(p, H, s);
    return launch_ce_bwd_t<DP, VQ_METRIC_DOT>(p, H, s);
}

bool aligned16(const void *p) { return ((uintptr_t)p & 15) == 0; }

int check_common(const vq_args *a) {
    if (!a) return fail(VQ_E_BADARG, "vq: null args");
    if (a->H <= 0 || a->Q <= 0 || a->M < 0 || a->K <= 0 || a->D <= 0) return fail(VQ_E_BADARG, "vq: non-positive size");
    if (a->metric != VQ_METRIC_EUCLID && a->metric != VQ_METRIC_DOT) return fail(VQ_E_BADARG, "vq: unknown metric");
    if (!a->x && a->M > 0) return fail(VQ_E_BADARG, "vq: x is null");
    return 0;
}

// Workspace layout: [keys: H*M int64][loss partials: floats]
long long ws_keys_bytes(int H, long long M) { return ((long long)H * M * 8 + 255) / 256 * 256; }
long long ws_loss_floats(int H, long long M, int Q) { return (long long)H * ((M + 31) / 32 + 8) * Q + (long long)H * 8192 + 64; }

void fill_search_params(SearchParams &p, const vq_args *a) {
    memset(&p, 0, sizeof(p));
    p.x = a->x; p.x_rs = a->x_rs; p.x_hs = a->x_hs;
    p.cb = a->cb; p.cb_hs = a->cb_hs; p.cb_qs = a->cb_qs;
    p.packed = a->packed; p.pk_hs = a->pk_hs; p.pk_qs = a->pk_qs;
    p.out = a->out; p.out_rs = a->out_rs; p.out_hs = a->out_hs;
    p.idx = (long long *)a->idx; p.idx_rs = a->idx_rs; p.idx_hs = a->idx_hs; p.idx_qs = a->idx_qs;
    p.best = a->best;
    p.M = a->M; p.K = a->K; p.D = a->D; p.Q = a->Q;
    {
        const int tc = kTileCodes * sub_tiles(padded_dim(a->D) ? padded_dim(a->D) : 256);
        p.ntiles = (a->K + tc - 1) / tc;
    }
    p.tiles_per_split = p.ntiles;
    p.pk_bytes = (unsigned)(vq_packed_floats(a->K, a->D) * 4);
    p.ste = (a->flags & VQ_F_STE) ? 1 : 0;
    p.vec_x = (a->D % 4 == 0 && a->x_rs % 4 == 0 && a->x_hs % 4 == 0 && aligned16(a->x)) ? 1 : 0;
    p.vec_fin = (p.vec_x && (!a->out || (a->out_rs % 4 == 0 && a->out_hs % 4 == 0 && aligned16(a->out))) &&
                 (!a->cb || (a->cb_hs % 4 == 0 && a->cb_qs % 4 == 0 && aligned16(a->cb)))) ? 1 : 0;
}

int run_finalize(const vq_args *a, const long long *keys, float *loss_part, hipStream_t s, int *nparts_out) {
    FinalizeParams f;
    memset(&f, 0, sizeof(f));
    f.keys = keys;
    f.x = a->x; f.x_rs = a->x_rs; f.x_hs = a->x_hs;
    f.cb = a->cb; f.cb_hs = a->cb_hs;
    f.out = a->out; f.out_rs = a->out_rs; f.out_hs = a->out_hs;
    f.idx = (long long *)a->idx; f.idx_rs = a->idx_rs; f.idx_hs = a->idx_hs;
    f.best = a->best;
    f.loss_part = loss_part;
    f.M = a->M; f.D = a->D; f.metric = a->metric; f.ste = (a->flags & VQ_F_STE) ? 1 : 0;
    long long blocks = (a->M + 3) / 4;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(vq_finalize_kernel, dim3((unsigned)blocks, (unsigned)a->H), dim3(256), 0, s, f);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "vq_finalize launch");
    if (nparts_out) *nparts_out = (int)(blocks * 4 * a->H);
    return 0;
}

int run_search_keys(const vq_args *a, long long idx_offset, long long *keys, hipStream_t s) {
    const int DP = padded_dim(a->D);
    const bool simple = (a->flags & VQ_F_FORCE_SIMPLE) || DP == 0;
    if (simple) {
        if (!a->cb) return fail(VQ_E_BADARG, "vq: natural codebook required for the scalar kernel");
        dim3 grid((unsigned)((a->M + 63) / 64), (unsigned)a->H);
        if (a->metric == VQ_METRIC_EUCLID)
            hipLaunchKernelGGL(vq_search_simple<VQ_METRIC_EUCLID>, grid, dim3(64), 0, s, a->x, a->x_rs, a->x_hs, a->cb,
                               a->cb_hs, a->M, a->K, a->D, idx_offset, keys);
        else
            hipLaunchKernelGGL(vq_search_simple<VQ_METRIC_DOT>, grid, dim3(64), 0, s, a->x, a->x_rs, a->x_hs, a->cb,
                               a->cb_hs, a->M, a->K, a->D, idx_offset, keys);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return hip_fail(e, "vq_search_simple launch");
        return 0;
    }
    if (!a->packed) return fail(VQ_E_BADARG, "vq: packed codebook is null");
    if (vq_packed_floats(a->K, a->D) * 4 >= (1ll << 31))
        return fail(VQ_E_UNSUPPORTED, "vq: packed codebook image >= 2 GiB (shard the codebook)");
    SearchParams p;
    fill_search_params(p, a);
    p.mode = kModeKeys;
    p.keys = keys;
    p.idx_offset = idx_offset;
    p.Q = 1;
    p.out = nullptr;
    p.loss_part = nullptr;
    const DevInfo &di = dev_info();
    const int cus = di.ok && di.cus > 0 ? di.cus : 256;
    int waves = (DP == 512) ? 4 : 8;
    long long wgs = (long long)a->H * ((a->M + 32 * waves - 1) / (32 * waves));
    if (waves == 8 && wgs < cus) {
        waves = 4;
        wgs = (long long)a->H * ((a->M + 127) / 128);
    }
    int splits = 1;
    if (wgs < 2 * cus) {
        splits = (int)((2 * cus + wgs - 1) / wgs);
        if (splits > p.ntiles) splits = p.ntiles;
        if (splits < 1) splits = 1;
    }
    p.tiles_per_split = (p.ntiles + splits - 1) / splits;
    splits = (p.ntiles + p.tiles_per_split - 1) / p.tiles_per_split;
    return launch_search(DP, waves, p, a->H, splits, a->metric, s);
}

}  // namespace

// =================================================================================================
// C ABI
// =================================================================================================
extern "C" {

const char *vq_last_error(void) { return g_err; }

#ifdef VQ_EXP_STAMPS
int vq_debug_read_stamps(unsigned long long *host, size_t n) {  // diagnostic build only
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_stamps), n * sizeof(unsigned long long));
}
#endif

int vq_device_info(char *buf, size_t n) {
    const DevInfo &di = dev_info();
    if (!di.ok) return fail(VQ_E_NODEVICE, "vq: no HIP device");
    snprintf(buf, n, "%s %d CUs", di.name, di.cus);
    return 0;
}

int64_t vq_packed_floats(int K, int D) {
    if (K <= 0 || D <= 0) return 0;
    const int DP = padded_dim(D);
    if (DP == 0) return 4;  // scalar kernel reads the natural codebook
    return (int64_t)round_up(K, kTileCodes * sub_tiles(DP)) * (DP + 4) + kPackSlack;
}

int64_t vq_workspace_bytes(int H, int64_t M, int Q) {
    if (H <= 0 || M < 0 || Q <= 0) return 0;
    return ws_keys_bytes(H, M) + ws_loss_floats(H, M, Q) * 4 + 256;
}

int vq_pack_codebooks_f32(const float *cb, int n_codebooks, int64_t cb_stride, int K, int D, int metric, float *packed,
                          void *stream) {
    if (!cb || !packed || n_codebooks <= 0 || K <= 0 || D <= 0) return fail(VQ_E_BADARG, "vq_pack: bad argument");
    if (metric != VQ_METRIC_EUCLID && metric != VQ_METRIC_DOT) return fail(VQ_E_BADARG, "vq_pack: unknown metric");
    const int DP = padded_dim(D);
    if (DP == 0) return 0;  // nothing to pack: the scalar kernel is used for D > 512
    if (!aligned16(packed)) return fail(VQ_E_BADARG, "vq_pack: packed buffer must be 16-byte aligned");
    const int Kp = round_up(K, kTileCodes * sub_tiles(DP));
    const long long pk_stride = vq_packed_floats(K, D);
    hipStream_t s = (hipStream_t)stream;
    // grid covers Kp rows plus at least one extra block whose threads zero the over-copy slack
    hipLaunchKernelGGL(vq_pack_kernel, dim3(Kp / 64 + 1, n_codebooks), dim3(64), 0, s, cb, (long long)cb_stride, K, Kp,
                       D, DP, metric, packed, pk_stride);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "vq_pack launch");
    return 0;
}

int vq_keys_init(int64_t *keys, int64_t n, void *stream) {
    if (!keys || n < 0) return fail(VQ_E_BADARG, "vq_keys_init: bad argument");
    if (n == 0) return 0;
    hipLaunchKernelGGL(vq_keys_init_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (long long *)keys, (long long)n);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "vq_keys_init launch");
    return 0;
}

int vq_search_keys_f32(const vq_args *a, int64_t idx_offset, int64_t *keys, void *stream) {
    int rc = check_common(a);
    if (rc) return rc;
    if (a->Q != 1) return fail(VQ_E_BADARG, "vq_search_keys: Q must be 1");
    if (!keys) return fail(VQ_E_BADARG, "vq_search_keys: keys is null");
    if (idx_offset < 0 || idx_offset + a->K > 0xFFFFFFFFll) return fail(VQ_E_BADARG, "vq_search_keys: index range");
    if (a->M == 0) return 0;
    return run_search_keys(a, idx_offset, (long long *)keys, (hipStream_t)stream);
}

int vq_finalize_keys_f32(const vq_args *a, const int64_t *keys, void *stream) {
    int rc = check_common(a);
    if (rc) return rc;
    if (a->Q != 1) return fail(VQ_E_BADARG, "vq_finalize_keys: Q must be 1");
    if (!keys || !a->cb) return fail(VQ_E_BADARG, "vq_finalize_keys: keys / cb is null");
    if (a->M == 0) {
        if (a->sq_err) hipMemsetAsync(a->sq_err, 0, sizeof(double), (hipStream_t)stream);
        return 0;
    }
    hipStream_t s = (hipStream_t)stream;
    float *loss_part = nullptr;
    if (a->sq_err) {
        if (!a->workspace || a->workspace_bytes < vq_workspace_bytes(a->H, a->M, 1))
            return fail(VQ_E_BADARG, "vq_finalize_keys: workspace too small");
        loss_part = (float *)((char *)a->workspace + ws_keys_bytes(a->H, a->M));
    }
    int nparts = 0;
    rc = run_finalize(a, (const long long *)keys, loss_part, s, &nparts);
    if (rc) return rc;
    if (a->sq_err) {
        hipLaunchKernelGGL(vq_loss_reduce_kernel, dim3(1), dim3(256), 0, s, loss_part, (long long)nparts, 1, a->sq_err);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return hip_fail(e, "vq_loss_reduce launch");
    }
    return 0;
}

int vq_quantize_f32(const vq_args *a, void *stream) {
    int rc = check_common(a);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    if (a->M > 0 && !a->idx) return fail(VQ_E_BADARG, "vq_quantize: idx is null");
    if (a->M > 0 && !a->cb) return fail(VQ_E_BADARG, "vq_quantize: natural codebook is null");
    if (a->M == 0) {
        if (a->sq_err) hipMemsetAsync(a->sq_err, 0, sizeof(double) * a->Q, s);
        return 0;
    }
    if ((a->sq_err || true) && (!a->workspace || a->workspace_bytes < vq_workspace_bytes(a->H, a->M, a->Q)))
        return fail(VQ_E_BADARG, "vq_quantize: workspace too small (see vq_workspace_bytes)");
    long long *keys = (long long *)a->workspace;
    float *loss_part = (float *)((char *)a->workspace + ws_keys_bytes(a->H, a->M));

    const int DP = padded_dim(a->D);
    const bool simple = (a->flags & VQ_F_FORCE_SIMPLE) || DP == 0;
    const DevInfo &di = dev_info();
    const int cus = di.ok && di.cus > 0 ? di.cus : 256;

    // ---- choose fused (one launch, no K split) or split (keys + finalize) ----
    bool fused = !simple;
    int waves = (DP == 512) ? 4 : 8;
    if (fused) {
        long long wgs = (long long)a->H * ((a->M + 32 * waves - 1) / (32 * waves));
        if (waves == 8 && wgs < cus) {
            waves = 4;
            wgs = (long long)a->H * ((a->M + 127) / 128);
        }
#ifdef VQ_EXP_WAVES
        waves = VQ_EXP_WAVES;  // diagnostic builds only
#endif
        const int ntiles = (a->K + kTileCodes * sub_tiles(DP) - 1) / (kTileCodes * sub_tiles(DP));
        // few workgroups and a long sweep: splitting K over workgroups fills the chip (Q == 1 only)
        if (a->Q == 1 && wgs * 2 <= cus && ntiles * sub_tiles(DP) >= 8 && ntiles >= 2) fused = false;
        if ((a->flags & VQ_F_FORCE_SPLIT) && a->Q == 1) fused = false;
    }

    if (fused) {
        if (!a->packed) return fail(VQ_E_BADARG, "vq_quantize: packed codebook is null");
        if (vq_packed_floats(a->K, a->D) * 4 >= (1ll << 31))
            return fail(VQ_E_UNSUPPORTED, "vq_quantize: packed codebook image >= 2 GiB (shard the codebook)");
        SearchParams p;
        fill_search_params(p, a);
        p.mode = kModeFused;
        p.loss_part = a->sq_err ? loss_part : nullptr;
        rc = launch_search(DP, waves, p, a->H, 1, a->metric, s);
        if (rc) return rc;
        if (a->sq_err) {
            const long long rows_per_wg = 32ll * waves;
            const long long nparts = (long long)a->H * ((a->M + rows_per_wg - 1) / rows_per_wg) * waves;
            hipLaunchKernelGGL(vq_loss_reduce_kernel, dim3(a->Q), dim3(256), 0, s, loss_part, nparts, a->Q, a->sq_err);
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) return hip_fail(e, "vq_loss_reduce launch");
        }
        return 0;
    }

    if (a->Q != 1) return fail(VQ_E_UNSUPPORTED, "vq_quantize: residual stages need the MFMA kernel (D <= 512)");
    rc = vq_keys_init((int64_t *)keys, (int64_t)a->H * a->M, stream);
    if (rc) return rc;
    rc = run_search_keys(a, 0, keys, s);
    if (rc) return rc;
    int nparts = 0;
    rc = run_finalize(a, keys, a->sq_err ? loss_part : nullptr, s, &nparts);
    if (rc) return rc;
    if (a->sq_err) {
        hipLaunchKernelGGL(vq_loss_reduce_kernel, dim3(1), dim3(256), 0, s, loss_part, (long long)nparts, 1, a->sq_err);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return hip_fail(e, "vq_loss_reduce launch");
    }
    return 0;
}

int vq_ema_accumulate_f32(const float *x, int64_t x_rs, int64_t x_hs, const int64_t *idx, int64_t idx_rs, int64_t idx_hs,
                          const uint8_t *mask, int H, int64_t M, int K, int D, float *counts, float *sums, void *stream) {
    if (H <= 0 || M < 0 || K <= 0 || D <= 0 || !counts || !sums) return fail(VQ_E_BADARG, "vq_ema_accumulate: bad argument");
    if (M == 0) return 0;
    if (!x || !idx) return fail(VQ_E_BADARG, "vq_ema_accumulate: null input");
    long long blocks = (M + 3) / 4;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(vq_ema_accumulate_kernel, dim3((unsigned)blocks, (unsigned)H), dim3(256), 0, (hipStream_t)stream, x,
                       (long long)x_rs, (long long)x_hs, (const long long *)idx, (long long)idx_rs, (long long)idx_hs, mask,
                       (long long)M, K, D, counts, sums);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "vq_ema_accumulate launch");
    return 0;
}

int vq_ema_update_f32(float *cluster_size, float *embed_avg, float *embeddings, const float *counts, const float *sums,
                      float *total_scratch, int H, int K, int D, float decay, float eps, int l2norm, void *stream) {
    if (!cluster_size || !embed_avg || !embeddings || !counts || !sums || !total_scratch || H <= 0 || K <= 0 || D <= 0)
        return fail(VQ_E_BADARG, "vq_ema_update: bad argument");
    hipStream_t s = (hipStream_t)stream;
    const float weight = 1.0f - decay;
    hipLaunchKernelGGL(vq_ema_sizes_kernel, dim3(H), dim3(256), 0, s, cluster_size, counts, K, weight, total_scratch);
    const long long rows = (long long)H * K;
    hipLaunchKernelGGL(vq_ema_codes_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, cluster_size, total_scratch,
                       embed_avg, sums, embeddings, H, K, D, weight, eps, l2norm);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "vq_ema_update launch");
    return 0;
}

static int fill_aux_params(AuxParams &p, const vq_args *a, const char *who) {
    memset(&p, 0, sizeof(p));
    if (!a->packed) return fail(VQ_E_BADARG, "vq: packed codebook is null");
    if (vq_packed_floats(a->K, a->D) * 4 >= (1ll << 31))
        return fail(VQ_E_UNSUPPORTED, "vq: packed codebook image >= 2 GiB (shard the codebook)");
    (void)who;
    p.x = a->x; p.x_rs = a->x_rs; p.x_hs = a->x_hs;
    p.packed = a->packed; p.pk_hs = a->pk_hs;
    p.pk_bytes = (unsigned)(vq_packed_floats(a->K, a->D) * 4);
    p.M = a->M; p.K = a->K; p.D = a->D;
    const int tc = kTileCodes * sub_tiles(padded_dim(a->D));
    p.ntiles = (a->K + tc - 1) / tc;
    p.vec_x = (a->D % 4 == 0 && a->x_rs % 4 == 0 && a->x_hs % 4 == 0 && aligned16(a->x)) ? 1 : 0;
    return 0;
}

int vq_similarities_f32(const vq_args *a, float *sims, int64_t sims_rs, int64_t sims_hs, void *stream) {
    int rc = check_common(a);
    if (rc) return rc;
    if (a->M == 0) return 0;
    if (!sims) return fail(VQ_E_BADARG, "vq_similarities: sims is null");
    hipStream_t s = (hipStream_t)stream;
    const int DP = padded_dim(a->D);
    if ((a->flags & VQ_F_FORCE_SIMPLE) || DP == 0) {
        if (!a->cb) return fail(VQ_E_BADARG, "vq_similarities: natural codebook required for the scalar kernel");
        const long long n = a->M * (long long)a->K;
        if ((n + 255) / 256 > 0x7FFFFFFFll) return fail(VQ_E_UNSUPPORTED, "vq_similarities: chunk too large for the scalar kernel");
        dim3 grid((unsigned)((n + 255) / 256), (unsigned)a->H);
        if (a->metric == VQ_METRIC_EUCLID)
            hipLaunchKernelGGL(vq_sims_simple<VQ_METRIC_EUCLID>, grid, dim3(256), 0, s, a->x, a->x_rs, a->x_hs, a->cb,
                               a->cb_hs, a->M, a->K, a->D, sims, (long long)sims_rs, (long long)sims_hs);
        else
            hipLaunchKernelGGL(vq_sims_simple<VQ_METRIC_DOT>, grid, dim3(256), 0, s, a->x, a->x_rs, a->x_hs, a->cb,
                               a->cb_hs, a->M, a->K, a->D, sims, (long long)sims_rs, (long long)sims_hs);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return hip_fail(e, "vq_sims_simple launch");
        return 0;
    }
    AuxParams p;
    rc = fill_aux_params(p, a, "vq_similarities");
    if (rc) return rc;
    p.sims = sims; p.sims_rs = sims_rs; p.sims_hs = sims_hs;
    p.vec_s = (a->K % 4 == 0 && sims_rs % 4 == 0 && sims_hs % 4 == 0 && aligned16(sims)) ? 1 : 0;
    return launch_aux(DP, p, a->H, a->metric, kAuxSims, s);
}

int vq_softmax_stats_f32(const vq_args *a, float scale, const int64_t *target, int64_t tgt_rs, int64_t tgt_hs, float *lse,
                         float *target_logit, void *stream) {
    int rc = check_common(a);
    if (rc) return rc;
    if (a->M == 0) return 0;
    if (!lse) return fail(VQ_E_BADARG, "vq_softmax_stats: lse is null");
    if (target && !target_logit) return fail(VQ_E_BADARG, "vq_softmax_stats: target_logit is null");
    const int DP = padded_dim(a->D);
    if (DP == 0) return fail(VQ_E_UNSUPPORTED, "vq_softmax_stats: D > 512 is not supported (use vq_similarities_f32 chunks)");
    AuxParams p;
    rc = fill_aux_params(p, a, "vq_softmax_stats");
    if (rc) return rc;
    p.scale = scale;
    p.target = (const long long *)target; p.tgt_rs = tgt_rs; p.tgt_hs = tgt_hs;
    p.lse = lse; p.tgt_logit = target_logit;
    return launch_aux(DP, p, a->H, a->metric, kAuxStats, (hipStream_t)stream);
}

int vq_ce_backward_f32(const vq_args *a, const float *lse, const float *target_logit, const int64_t *target, int64_t tgt_rs,
                       int64_t tgt_hs, const float *coef, float *grad_x, int64_t gx_rs, int64_t gx_hs, void *stream) {
    int rc = check_common(a);
    if (rc) return rc;
    if (a->M == 0) return 0;
    if (!lse || !target_logit || !target || !coef || !grad_x || !a->cb)
        return fail(VQ_E_BADARG, "vq_ce_backward: null argument (lse / target_logit / target / coef / grad_x / cb)");
    const int DP = padded_dim(a->D);
    if (DP == 0) return fail(VQ_E_UNSUPPORTED, "vq_ce_backward: D > 512 (use vq_similarities_f32 row chunks)");
    AuxParams ap;
    rc = fill_aux_params(ap, a, "vq_ce_backward");
    if (rc) return rc;
    CeBwdParams p;
    memset(&p, 0, sizeof(p));
    p.x = ap.x; p.x_rs = ap.x_rs; p.x_hs = ap.x_hs;
    p.packed = ap.packed; p.pk_hs = ap.pk_hs; p.pk_bytes = ap.pk_bytes;
    p.M = ap.M; p.K = ap.K; p.D = ap.D; p.ntiles = ap.ntiles; p.vec_x = ap.vec_x;
    p.lse = lse;
    p.target = (const long long *)target; p.tgt_rs = tgt_rs; p.tgt_hs = tgt_hs;
    p.coef = coef;
    p.cb = a->cb; p.cb_hs = a->cb_hs;
    p.tgt_logit = target_logit;
    p.gx = grad_x; p.gx_rs = gx_rs; p.gx_hs = gx_hs;
    hipStream_t s = (hipStream_t)stream;
    switch (DP) {
        case 32: return launch_ce_bwd_m<32>(p, a->H, a->metric, s);
        case 64: return launch_ce_bwd_m<64>(p, a->H, a->metric, s);
        case 128: return launch_ce_bwd_m<128>(p, a->H, a->metric, s);
        case 256: return launch_ce_bwd_m<256>(p, a->H, a->metric, s);
        case 512: return launch_ce_bwd_m<512>(p, a->H, a->metric, s);
    }
    return fail(VQ_E_UNSUPPORTED, "vq_ce_backward: unsupported padded dim");
}

int vq_nearest_f32(const vq_args *a, void *stream) {
    if (a && a->Q != 1) return fail(VQ_E_BADARG, "vq_nearest_f32: Q must be 1");
    return vq_quantize_f32(a, stream);
}

int vq_residual_f32(const vq_args *a, void *stream) { return vq_quantize_f32(a, stream); }

}  // extern "C"
