// vq_kernels.hip -- nearest-codebook search for MI355X (gfx950 / CDNA4).  Hand-written HIP, no
// compatibility layers.  See DESIGN.md for the full description; summary of the data path:
//
//   pack      natural codebook [K, D]  ->  packed image  [Kp][Dp + 4]  (even/odd de-interleave inside
//             each group of 8 dims, pre-scaled by -2 for Euclid, |c|^2 in float Dp of every row)
//             (+ a "some code is non-finite" word per image: ATen's argmax rule for NaN / inf inputs is restored out of line,
//             see repair_nonfinite_rows in vq_search.inc)
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
// Source layout (ONE source file: the .inc files are included below inside the anonymous namespace, in this order; the file
// is compiled either whole or once per build part -- see "Build parts"):
//   vq_common.inc        constants, error strings, padded-dim table, packed (value, index) keys
//   vq_pack.inc          natural codebook -> packed image
//   vq_search.inc        the hot kernel (tile geometry, LDS-DMA staging, MFMA fragment pipeline, tie-exact epilogue, finalize)
//   vq_search_pair.inc   the same search for 256 < D <= 512 with the dims split over a pair of waves (accumulator hand-off)
//   vq_search_persist.inc  inference search at Dp = 256 with block b's gather hidden inside block b + 1's sweep
//   vq_search_resident.inc small codebooks: the packed image stays in LDS, no barriers, rows streamed past it by LDS-DMA slabs
//   vq_similarity.inc    the same sweep with the similarity / online-softmax epilogues, fused cross-entropy backward
//   vq_finalize_ema.inc  scalar fallback search, finalize-from-keys, loss reduction, EMA codebook update
//   this file            host-side dispatch and the C ABI (include/vq_mi355x.h)
//
// Reference lines replaced (relative to the reference root): vector_quantization/codebooks.py:386-397,
// utils/general.py:126-136,159-163, vector_quantize_pytorch.py:261-279,361-364, residual_vq.py:212-243.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/vq_mi355x.h"

// Build parts.  `hipcc ... vq_kernels.hip` (VQ_PART undefined) builds everything as ONE translation unit.  build.sh compiles
// the same file once per part (-DVQ_PART=n, in parallel) and links the objects: every part sees the same templates, but only
// its own launchers are defined -- and with them instantiated -- there; the other parts call them through the
// vqi::part_* entry points declared below.
//   0 C ABI, planners, small kernels (pack, scalar search, finalize, EMA)      4 search Dp = 512 + wave-pair kernel
//   1 search Dp = 32 / 64         2 search Dp = 128                             5 similarity / softmax-statistics sweeps
//   3 search Dp = 256 + persistent kernel + full slices of wide rows            6 fused cross-entropy backward
#ifndef VQ_PART
#define VQ_PART -1
#endif
#define VQ_OWN(part) (VQ_PART < 0 || VQ_PART == (part))

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

#include "vq_common.inc"
#if VQ_OWN(0)
#include "vq_pack.inc"
#endif
#include "vq_search.inc"
#include "vq_search_pair.inc"
#include "vq_search_persist.inc"
#include "vq_search_resident.inc"
#include "vq_similarity.inc"
#if VQ_OWN(0)
#include "vq_finalize_ema.inc"
#endif

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
struct DevInfo {
    int cus = 0;
    int clock_mhz = 2400;  // shader clock (hipDeviceProp_t::clockRate), what the launch plans price a sub-tile with
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
            if (prop.clockRate > 0) info.clock_mhz = prop.clockRate / 1000;
            snprintf(info.name, sizeof(info.name), "%s", prop.gcnArchName);
            info.ok = true;
            cached_dev = dev;
        } else {
            info.ok = false;
        }
    }
    return info;
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a per-device property of a kernel: one flag per device ordinal
// (per thread and per instantiation), so a process that drives several GPUs raises the limit on each of them.
constexpr int kMaxDevices = 64;

template <typename K>
int allow_big_lds(K kern, bool (&done)[kMaxDevices]) {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) return fail(VQ_E_NODEVICE, "vq: no HIP device");
    const bool tracked = dev >= 0 && dev < kMaxDevices;
    if (tracked && done[dev]) return 0;
    hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute");
    if (tracked) done[dev] = true;
    return 0;
}

template <int DP, int WAVES, int METRIC, int MULTI, bool LSE = false, int XT = 0, int WIDE = 0>
int launch_search_t(const SearchParams &p, int H, int splits, hipStream_t s) {
    using G = Geo<DP, WAVES>;
    const size_t lds = (size_t)(MULTI ? G::MAIN_FLOATS_M : G::MAIN_FLOATS_S) * 4 + (size_t)WAVES * p.Q * 32 * 4 +
                       ((MULTI && p.loss_part) ? (size_t)WAVES * p.Q * 64 * 4 : 0);
    if (lds > 160 * 1024) return fail(VQ_E_UNSUPPORTED, "vq_search: LDS budget exceeded (too many residual stages)");
    auto kern = vq_search_mfma<DP, WAVES, METRIC, MULTI, LSE, XT, WIDE>;
    static thread_local bool attr_done[kMaxDevices] = {};
    if (int rc = allow_big_lds(kern, attr_done)) return rc;
    const long long rows_per_wg = 32ll * WAVES;
    dim3 grid((unsigned)((p.M + rows_per_wg - 1) / rows_per_wg), (unsigned)H, (unsigned)splits);
    hipLaunchKernelGGL(kern, grid, dim3(WAVES * 64), lds, s, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "vq_search_mfma launch");
    return 0;
}

// LDS bytes of a residual (multi-stage) launch with Q stages; the budget is the CU's 160 KiB
template <int DP, int WAVES>
constexpr size_t multi_lds_bytes(int Q, bool with_loss) {
    return (size_t)Geo<DP, WAVES>::MAIN_FLOATS_M * 4 + (size_t)WAVES * Q * 32 * 4 + (with_loss ? (size_t)WAVES * Q * 64 * 4 : 0);
}

template <int DP, int WAVES>
int max_stages_t(bool with_loss) {
    int q = 1;
    while (q < 4096 && multi_lds_bytes<DP, WAVES>(q + 1, with_loss) <= 160 * 1024) ++q;
    return q;
}

template <int DP, int WAVES>
int launch_search_m(const SearchParams &p, int H, int splits, int metric, hipStream_t s) {
    if (p.Q > 1) {  // residual stages: eval (1) and straight-through (2) arithmetic are separate instantiations
        if (p.ste) {
            if (metric == VQ_METRIC_EUCLID) return launch_search_t<DP, WAVES, VQ_METRIC_EUCLID, 2>(p, H, splits, s);
            return launch_search_t<DP, WAVES, VQ_METRIC_DOT, 2>(p, H, splits, s);
        }
        if (metric == VQ_METRIC_EUCLID) return launch_search_t<DP, WAVES, VQ_METRIC_EUCLID, 1>(p, H, splits, s);
        return launch_search_t<DP, WAVES, VQ_METRIC_DOT, 1>(p, H, splits, s);
    }
    if (p.xt == 1) {  // fp16 rows, widened in the prologue (inference)
        if (metric == VQ_METRIC_EUCLID) return launch_search_t<DP, WAVES, VQ_METRIC_EUCLID, 0, false, 1>(p, H, splits, s);
        return launch_search_t<DP, WAVES, VQ_METRIC_DOT, 0, false, 1>(p, H, splits, s);
    }
    if (p.xt == 2) {  // bf16 rows
        if (metric == VQ_METRIC_EUCLID) return launch_search_t<DP, WAVES, VQ_METRIC_EUCLID, 0, false, 2>(p, H, splits, s);
        return launch_search_t<DP, WAVES, VQ_METRIC_DOT, 0, false, 2>(p, H, splits, s);
    }
    if (p.lse) {  // search + log-sum-exp in one sweep (cross-entropy commitment loss)
        if (metric == VQ_METRIC_EUCLID) return launch_search_t<DP, WAVES, VQ_METRIC_EUCLID, 0, true>(p, H, splits, s);
        return launch_search_t<DP, WAVES, VQ_METRIC_DOT, 0, true>(p, H, splits, s);
    }
    if (metric == VQ_METRIC_EUCLID) return launch_search_t<DP, WAVES, VQ_METRIC_EUCLID, 0>(p, H, splits, s);
    return launch_search_t<DP, WAVES, VQ_METRIC_DOT, 0>(p, H, splits, s);
}

// VQ_SINGLE_WAVE_512=1 in the environment selects the one-wave-per-row-block kernel for D > 256 (A/B measurements)
bool use_pair512() {
    static const bool off = getenv("VQ_SINGLE_WAVE_512") != nullptr;
    return !off;
}

// The wave-pair kernel runs wave B one tile behind wave A: one extra step per sweep.  Worth it from 8 tiles per sweep on.
// Residual stacks (round 3) as long as the winners of every stage fit its LDS (Q <= 26); VQ_PAIR_NO_MULTI=1: one-wave kernel (A/B).
bool pair_selected(int DP, int Q, int tiles_per_sweep) {
    if (!(DP == 512 && tiles_per_sweep >= 8 && use_pair512())) return false;
    if (Q == 1) return true;
    return getenv("VQ_PAIR_NO_MULTI") == nullptr && Q <= PairGeo::max_stages();  // (read per call: tests compare both kernels in one process)
}

// 256 < D <= 512: dims split over wave pairs (vq_search_pair.inc); 8 waves = 4 pairs = 128 rows per workgroup
template <int METRIC, bool LSE = false, int XT = 0, int WIDE = 0, int MULTI = 0>
int launch_pair_t(const SearchParams &p, int H, int splits, hipStream_t s) {
    const size_t lds = PairGeo::lds_bytes(MULTI ? p.Q : 1);
    auto kern = vq_search_pair512<METRIC, LSE, XT, WIDE, MULTI>;
    static thread_local bool attr_done[kMaxDevices] = {};
    if (int rc = allow_big_lds(kern, attr_done)) return rc;
    dim3 grid((unsigned)((p.M + 127) / 128), (unsigned)H, (unsigned)splits);
    hipLaunchKernelGGL(kern, grid, dim3(512), lds, s, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "vq_search_pair512 launch");
    return 0;
}

#if VQ_OWN(4)  // (not a template: defining it instantiates the wave-pair kernels)
int launch_pair_any(const SearchParams &p, int H, int splits, int metric, hipStream_t s) {
    const bool eu = metric == VQ_METRIC_EUCLID;
    if (p.Q > 1) {  // residual stacks: eval (1) and straight-through (2) arithmetic, as in launch_search_m
        if (p.ste) return eu ? launch_pair_t<VQ_METRIC_EUCLID, false, 0, 0, 2>(p, H, splits, s) : launch_pair_t<VQ_METRIC_DOT, false, 0, 0, 2>(p, H, splits, s);
        return eu ? launch_pair_t<VQ_METRIC_EUCLID, false, 0, 0, 1>(p, H, splits, s) : launch_pair_t<VQ_METRIC_DOT, false, 0, 0, 1>(p, H, splits, s);
    }
    if (p.xt == 1) return eu ? launch_pair_t<VQ_METRIC_EUCLID, false, 1>(p, H, splits, s) : launch_pair_t<VQ_METRIC_DOT, false, 1>(p, H, splits, s);
    if (p.xt == 2) return eu ? launch_pair_t<VQ_METRIC_EUCLID, false, 2>(p, H, splits, s) : launch_pair_t<VQ_METRIC_DOT, false, 2>(p, H, splits, s);
    if (p.lse) return eu ? launch_pair_t<VQ_METRIC_EUCLID, true>(p, H, splits, s) : launch_pair_t<VQ_METRIC_DOT, true>(p, H, splits, s);
    return eu ? launch_pair_t<VQ_METRIC_EUCLID>(p, H, splits, s) : launch_pair_t<VQ_METRIC_DOT>(p, H, splits, s);
}
#endif

// workgroups per head of the persistent kernel: one 8-wave workgroup per CU, the CUs shared by the heads
inline long long persist_grid_x(long long M, int H, int cus) {
    const long long nblk = (M + 255) / 256;
    long long gx = cus / H;
    if (gx < 1) gx = 1;
    return gx > nblk ? nblk : gx;
}

template <int METRIC, bool TRAIN = false>
int launch_persist_t(const SearchParams &p, int H, int cus, hipStream_t s) {
    using G = Geo<256, 8>;
    const size_t lds = (size_t)G::MAIN_FLOATS_S * 4 + 2 * 8 * 32 * 4;
    auto kern = vq_search_persist<256, 8, METRIC, TRAIN>;
    static thread_local bool attr_done[kMaxDevices] = {};
    if (int rc = allow_big_lds(kern, attr_done)) return rc;
    const long long gx = persist_grid_x(p.M, H, cus);
    hipLaunchKernelGGL(kern, dim3((unsigned)gx, (unsigned)H, 1), dim3(512), lds, s, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "vq_search_persist launch");
    return 0;
}

// small codebooks: the image stays in LDS, one 8-wave workgroup per CU, waves walk 32-row blocks (vq_search_resident.inc)
template <int DP, int METRIC>
int launch_resident_t(const SearchParams &p, int H, int cus, hipStream_t s) {
    const size_t lds = ResGeo<DP>::lds_bytes(p.res_img_floats, p.res_nbuf);
    const long long nwb = (p.M + 31) / 32;
    long long gx = cus / H;
    if (gx < 1) gx = 1;
    if (gx > (nwb + 7) / 8) gx = (nwb + 7) / 8;
    if constexpr (DP < 128) {
        if (p.res_nbuf > 1) {  // a slab buffer per slab of a block
            auto kern = vq_search_resident<DP, METRIC, true>;
            static thread_local bool attr_done_r[kMaxDevices] = {};
            if (int rc = allow_big_lds(kern, attr_done_r)) return rc;
            hipLaunchKernelGGL(kern, dim3((unsigned)gx, (unsigned)H, 1), dim3(512), lds, s, p);
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) return hip_fail(e, "vq_search_resident launch");
            return 0;
        }
    }
    auto kern = vq_search_resident<DP, METRIC, false>;
    static thread_local bool attr_done[kMaxDevices] = {};
    if (int rc = allow_big_lds(kern, attr_done)) return rc;
    hipLaunchKernelGGL(kern, dim3((unsigned)gx, (unsigned)H, 1), dim3(512), lds, s, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "vq_search_resident launch");
    return 0;
}

template <int DP>
int launch_resident_m(const SearchParams &p, int H, int cus, int metric, hipStream_t s) {
    return metric == VQ_METRIC_EUCLID ? launch_resident_t<DP, VQ_METRIC_EUCLID>(p, H, cus, s) : launch_resident_t<DP, VQ_METRIC_DOT>(p, H, cus, s);
}

template <int DP, int WAVES, int METRIC, int MODE>
int launch_aux_t(const AuxParams &p, int H, hipStream_t s) {
    using G = Geo<DP, WAVES>;
    const size_t lds = (size_t)G::MAIN_FLOATS * 4;
    auto kern = vq_sweep_aux<DP, WAVES, METRIC, MODE>;
    static thread_local bool attr_done[kMaxDevices] = {};
    if (int rc = allow_big_lds(kern, attr_done)) return rc;
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

template <int DP, int METRIC>
int launch_ce_bwd_t(const CeBwdParams &p, int H, hipStream_t s) {
    using G = Geo<DP, 4>;
    const size_t stage_floats = (size_t)4 * (32 * CeGeo<DP>::GS + 96);
    const size_t lds = 4 * ((size_t)G::MAIN_FLOATS > stage_floats ? (size_t)G::MAIN_FLOATS : stage_floats);
    auto kern = vq_ce_backward<DP, METRIC>;
    static thread_local bool attr_done[kMaxDevices] = {};
    if (int rc = allow_big_lds(kern, attr_done)) return rc;
    dim3 grid((unsigned)((p.M + 127) / 128), (unsigned)H, (unsigned)CeGeo<DP>::NH);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "vq_ce_backward launch");
    return 0;
}

// Dp = 256: the two contractions on a pair of waves (two waves per SIMD); VQ_CE_NO_ROLES=1 keeps the one-wave kernel (A/B runs)
template <int METRIC>
int launch_ce_bwd_roles_t(const CeBwdParams &p, int H, hipStream_t s) {
    const size_t lds = (size_t)CeRolesGeo::LDS_F * 4;
    auto kern = vq_ce_backward_roles<METRIC>;
    static thread_local bool attr_done[kMaxDevices] = {};
    if (int rc = allow_big_lds(kern, attr_done)) return rc;
    dim3 grid((unsigned)((p.M + 32 * CeRolesGeo::NB - 1) / (32 * CeRolesGeo::NB)), (unsigned)H, 1);
    hipLaunchKernelGGL(kern, grid, dim3(512), lds, s, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "vq_ce_backward_roles launch");
    return 0;
}

// Dp = 512: four roles per row block (S cut in two, G in two halves of the positions), 64 rows per workgroup
template <int METRIC>
int launch_ce_bwd_roles512_t(const CeBwdParams &p, int H, hipStream_t s) {
    const size_t lds = (size_t)CeRoles512Geo::LDS_F * 4;
    auto kern = vq_ce_backward_roles512<METRIC>;
    static thread_local bool attr_done[kMaxDevices] = {};
    if (int rc = allow_big_lds(kern, attr_done)) return rc;
    dim3 grid((unsigned)((p.M + 32 * CeRoles512Geo::NQ - 1) / (32 * CeRoles512Geo::NQ)), (unsigned)H, 1);
    hipLaunchKernelGGL(kern, grid, dim3(512), lds, s, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "vq_ce_backward_roles512 launch");
    return 0;
}

template <int DP>
int launch_ce_bwd_m(const CeBwdParams &p, int H, int metric, hipStream_t s) {
    if constexpr (DP == 512) {
        if (getenv("VQ_CE_NO_ROLES") == nullptr) {
            if (metric == VQ_METRIC_EUCLID) return launch_ce_bwd_roles512_t<VQ_METRIC_EUCLID>(p, H, s);
            return launch_ce_bwd_roles512_t<VQ_METRIC_DOT>(p, H, s);
        }
    }
    if constexpr (DP == 256) {
        if (getenv("VQ_CE_NO_ROLES") == nullptr) {  // (read per call: tests switch between the two kernels in one process)
            if (metric == VQ_METRIC_EUCLID) return launch_ce_bwd_roles_t<VQ_METRIC_EUCLID>(p, H, s);
            return launch_ce_bwd_roles_t<VQ_METRIC_DOT>(p, H, s);
        }
    }
    if (metric == VQ_METRIC_EUCLID) return launch_ce_bwd_t<DP, VQ_METRIC_EUCLID>(p, H, s);
    return launch_ce_bwd_t<DP, VQ_METRIC_DOT>(p, H, s);
}

#ifndef VQ_EXP_RESIDENT_MIN_ROWS_PER_CU
#define VQ_EXP_RESIDENT_MIN_ROWS_PER_CU 512  // rows per CU from which the resident-codebook kernel takes a small codebook
#endif
#ifndef VQ_EXP_WIDE_SLICE
#define VQ_EXP_WIDE_SLICE 512  // dims per slice of rows wider than 512 dims (256: the round-2 scheme, twice the accumulator traffic)
#endif
constexpr int kWideSlice = VQ_EXP_WIDE_SLICE;                // dims per slice: 512 (4-wave workgroups) or 256 (8-wave)
constexpr int kWideWaves = kWideSlice == 512 ? 4 : 8;
constexpr int kWideRows = 32 * kWideWaves;                   // rows per workgroup
#ifndef VQ_EXP_WIDE_CHUNK_MB
#define VQ_EXP_WIDE_CHUNK_MB 512
#endif
constexpr long long kWideChunkBytes = (long long)VQ_EXP_WIDE_CHUNK_MB << 20;  // accumulator workspace per (row chunk, code chunk)
constexpr int kWideCodes = 4096;                    // codes per chunk

template <int DP, int WIDE>
int launch_wide_t(const SearchParams &p, int H, int splits, int metric, hipStream_t s) {
    if (metric == VQ_METRIC_EUCLID) return launch_search_t<DP, kWideWaves, VQ_METRIC_EUCLID, 0, false, 0, WIDE>(p, H, splits, s);
    return launch_search_t<DP, kWideWaves, VQ_METRIC_DOT, 0, false, 0, WIDE>(p, H, splits, s);
}

// one slice of rows wider than 512 dims: WIDE = 1 a full slice (Dp = the slice width only), 2 / 3 the last slice
template <int DP>
int launch_wide_any(int wide, const SearchParams &p, int H, int splits, int metric, hipStream_t s) {
    if (wide == 1) {
        if constexpr (DP >= 256) return launch_wide_t<DP, 1>(p, H, splits, metric, s);
        return fail(VQ_E_UNSUPPORTED, "vq_search: a full slice of wide rows is 256 (or 512) dims");
    }
    return wide == 3 ? launch_wide_t<DP, 3>(p, H, splits, metric, s) : launch_wide_t<DP, 2>(p, H, splits, metric, s);
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// entry points of the build parts: declared everywhere, defined (= their kernels instantiated) in the owning part only
// ------------------------------------------------------------------------------------------------
namespace vqi {
template <int DP> int part_search(int waves, const SearchParams &p, int H, int splits, int metric, hipStream_t s);
template <int DP> int part_wide(int wide, const SearchParams &p, int H, int splits, int metric, hipStream_t s);
template <int DP> int part_resident(const SearchParams &p, int H, int cus, int metric, hipStream_t s);
template <int DP> int part_aux(const AuxParams &p, int H, int metric, int mode, hipStream_t s);
template <int DP> int part_ce_bwd(const CeBwdParams &p, int H, int metric, hipStream_t s);
int part_pair(const SearchParams &p, int H, int splits, int metric, hipStream_t s);
int part_persist(const SearchParams &p, int H, int cus, int metric, hipStream_t s);
#define VQ_DECLARE_PARTS(DP)                                                                                         \
    template <> int part_search<DP>(int waves, const SearchParams &p, int H, int splits, int metric, hipStream_t s); \
    template <> int part_wide<DP>(int wide, const SearchParams &p, int H, int splits, int metric, hipStream_t s);    \
    template <> int part_aux<DP>(const AuxParams &p, int H, int metric, int mode, hipStream_t s);                    \
    template <> int part_ce_bwd<DP>(const CeBwdParams &p, int H, int metric, hipStream_t s);
template <> int part_resident<32>(const SearchParams &p, int H, int cus, int metric, hipStream_t s);
template <> int part_resident<64>(const SearchParams &p, int H, int cus, int metric, hipStream_t s);
template <> int part_resident<128>(const SearchParams &p, int H, int cus, int metric, hipStream_t s);
VQ_DECLARE_PARTS(32)
VQ_DECLARE_PARTS(64)
VQ_DECLARE_PARTS(128)
VQ_DECLARE_PARTS(256)
VQ_DECLARE_PARTS(512)
#undef VQ_DECLARE_PARTS

#define VQ_DEFINE_SEARCH_PART(DP, W4)                                                                                 \
    template <> int part_search<DP>(int waves, const SearchParams &p, int H, int splits, int metric, hipStream_t s) { \
        return waves == 8 ? launch_search_m<DP, 8>(p, H, splits, metric, s) : launch_search_m<DP, W4>(p, H, splits, metric, s); \
    }                                                                                                                 \
    template <> int part_wide<DP>(int wide, const SearchParams &p, int H, int splits, int metric, hipStream_t s) {    \
        return launch_wide_any<DP>(wide, p, H, splits, metric, s);                                                    \
    }
#define VQ_DEFINE_RESIDENT_PART(DP) \
    template <> int part_resident<DP>(const SearchParams &p, int H, int cus, int metric, hipStream_t s) { return launch_resident_m<DP>(p, H, cus, metric, s); }
#if VQ_OWN(1)
VQ_DEFINE_SEARCH_PART(32, 4)
VQ_DEFINE_SEARCH_PART(64, 4)
VQ_DEFINE_RESIDENT_PART(32)
VQ_DEFINE_RESIDENT_PART(64)
#endif
#if VQ_OWN(2)
VQ_DEFINE_SEARCH_PART(128, 4)
VQ_DEFINE_RESIDENT_PART(128)
#endif
#undef VQ_DEFINE_RESIDENT_PART
#if VQ_OWN(3)
VQ_DEFINE_SEARCH_PART(256, 4)
int part_persist(const SearchParams &p, int H, int cus, int metric, hipStream_t s) {
    if (p.ste || p.loss_part)  // training-mode call: the deferred copy does the straight-through / squared-error arithmetic
        return metric == VQ_METRIC_EUCLID ? launch_persist_t<VQ_METRIC_EUCLID, true>(p, H, cus, s) : launch_persist_t<VQ_METRIC_DOT, true>(p, H, cus, s);
    return metric == VQ_METRIC_EUCLID ? launch_persist_t<VQ_METRIC_EUCLID>(p, H, cus, s) : launch_persist_t<VQ_METRIC_DOT>(p, H, cus, s);
}
#endif
#if VQ_OWN(4)
template <> int part_search<512>(int, const SearchParams &p, int H, int splits, int metric, hipStream_t s) {
    return launch_search_m<512, 4>(p, H, splits, metric, s);
}
template <> int part_wide<512>(int wide, const SearchParams &p, int H, int splits, int metric, hipStream_t s) {
#if VQ_EXP_WIDE_SLICE == 512
    // a 512-dim slice of wider rows: the wave-pair kernel (two waves per SIMD, accumulator hand-off) when the sweep is long
    // enough for its extra pipeline step, else the one-wave kernel
    if (pair_selected(512, 1, p.tiles_per_split)) {
        const bool eu = metric == VQ_METRIC_EUCLID;
        if (wide == 1) return eu ? launch_pair_t<VQ_METRIC_EUCLID, false, 0, 1>(p, H, splits, s) : launch_pair_t<VQ_METRIC_DOT, false, 0, 1>(p, H, splits, s);
        if (wide == 3) return eu ? launch_pair_t<VQ_METRIC_EUCLID, false, 0, 3>(p, H, splits, s) : launch_pair_t<VQ_METRIC_DOT, false, 0, 3>(p, H, splits, s);
        return eu ? launch_pair_t<VQ_METRIC_EUCLID, false, 0, 2>(p, H, splits, s) : launch_pair_t<VQ_METRIC_DOT, false, 0, 2>(p, H, splits, s);
    }
    return launch_wide_any<512>(wide, p, H, splits, metric, s);
#else
    (void)wide; (void)p; (void)H; (void)splits; (void)metric; (void)s;
    return fail(VQ_E_UNSUPPORTED, "vq_search: unsupported padded dim");
#endif
}
int part_pair(const SearchParams &p, int H, int splits, int metric, hipStream_t s) { return launch_pair_any(p, H, splits, metric, s); }
#endif
#undef VQ_DEFINE_SEARCH_PART
#if VQ_OWN(5)
#define VQ_DEFINE_AUX_PART(DP) \
    template <> int part_aux<DP>(const AuxParams &p, int H, int metric, int mode, hipStream_t s) { return launch_aux_m<DP, 4>(p, H, metric, mode, s); }
VQ_DEFINE_AUX_PART(32)
VQ_DEFINE_AUX_PART(64)
VQ_DEFINE_AUX_PART(128)
VQ_DEFINE_AUX_PART(256)
VQ_DEFINE_AUX_PART(512)
#undef VQ_DEFINE_AUX_PART
#endif
#if VQ_OWN(6)
#define VQ_DEFINE_CE_PART(DP) \
    template <> int part_ce_bwd<DP>(const CeBwdParams &p, int H, int metric, hipStream_t s) { return launch_ce_bwd_m<DP>(p, H, metric, s); }
VQ_DEFINE_CE_PART(32)
VQ_DEFINE_CE_PART(64)
VQ_DEFINE_CE_PART(128)
VQ_DEFINE_CE_PART(256)
VQ_DEFINE_CE_PART(512)
#undef VQ_DEFINE_CE_PART
#endif
}  // namespace vqi

#if VQ_OWN(0)
namespace vqi {
thread_local char g_err[512] = "";
}
namespace {

// Plain inference call at Dp = 256 (one stage, no straight-through, no loss, aligned fp32 rows, >= 32 sub-tiles per sweep,
// several row blocks per CU): persistent workgroups that copy block b's winners during block b + 1's sweep.
// VQ_NO_PERSIST=1 in the environment keeps the one-block-per-workgroup kernel (A/B measurements).
bool persist_selected(int DP, int waves, const SearchParams &p, int H, int splits, int cus) {
    static const bool off = getenv("VQ_NO_PERSIST") != nullptr;
    if (off || DP != 256 || waves != 8 || splits != 1 || p.Q != 1 || p.mode != kModeFused) return false;
    if (p.lse || p.xt || !p.vec_x || !p.vec_fin || p.D % 4) return false;
    if (!p.out && !p.loss_part) return false;  // (nothing to copy)
    if ((p.ste || p.loss_part) && getenv("VQ_NO_PERSIST_TRAIN") != nullptr) return false;  // (A/B and tests: training-mode calls on the one-block kernel; read per call)
    const int nsub = p.ntiles * sub_tiles(DP);
    if (nsub < 32 || nsub > 96) return false;  // one row per sub-tile needs 32; beyond ~100 the finalize is < 1 % of a block
    const long long nblk = (p.M + 32 * waves - 1) / (32 * waves);
    return nblk * H >= 2ll * cus;  // at least two blocks per resident workgroup
}

bool aligned16(const void *p) { return ((uintptr_t)p & 15) == 0; }

// Small codebook, plain inference call (one stage, no straight-through / loss / LSE, aligned fp32 rows of D % 16 == 0 dims,
// Dp <= 128) whose image fits the LDS beside the row slabs, and enough rows to give every wave slot of the chip a few 32-row
// blocks: the resident-codebook kernel.  VQ_NO_RESIDENT=1 in the environment keeps the tile-streaming kernels (A/B measurements).
int resident_image_for(const vq_args *a, int DP, bool lse, int cus) {
    static const bool off = getenv("VQ_NO_RESIDENT") != nullptr;
    if (off || DP == 0 || DP > 128 || a->Q != 1 || lse || a->sq_err || !a->out || !a->idx || !a->cb || !a->packed) return 0;
    if (a->flags & (VQ_F_STE | VQ_F_FORCE_SIMPLE | VQ_F_FORCE_SPLIT | VQ_F_X_F16 | VQ_F_X_BF16)) return 0;
    if (a->D % 16 || a->x_rs % 4 || a->x_hs % 4 || !aligned16(a->x)) return 0;
    if (a->out_rs % 4 || a->out_hs % 4 || !aligned16(a->out) || a->cb_hs % 4 || !aligned16(a->cb)) return 0;
    const int nsub_k = (a->K + kTileCodes - 1) / kTileCodes;
    if (nsub_k < DP / 16) return 0;  // (the sweep's first Dp / 16 sub-tiles carry the next block's slabs)
    if (nsub_k > 8) return 0;        // measured (gpurun_out/r3/t5_res_ab.log): from K = 512 on the tile-streaming kernels are ahead again
    const int img = resident_image_floats(a->K, DP);
    const size_t lds = ((size_t)img + 8 * 512) * 4 + 2 * 8 * 32 * 4;
    if (lds > 160 * 1024) return 0;
    if ((long long)a->H * a->M < (long long)VQ_EXP_RESIDENT_MIN_ROWS_PER_CU * cus) return 0;
    return img;
}

int launch_search(int DP, int waves, const SearchParams &p, int H, int splits, int metric, hipStream_t s) {
    {
        const DevInfo &di = dev_info();
        const int cus = di.ok && di.cus > 0 ? di.cus : 256;
        if (p.res_img_floats > 0 && splits == 1 && p.mode == kModeFused) {
            switch (DP) {
                case 32: return vqi::part_resident<32>(p, H, cus, metric, s);
                case 64: return vqi::part_resident<64>(p, H, cus, metric, s);
                case 128: return vqi::part_resident<128>(p, H, cus, metric, s);
            }
        }
        if (persist_selected(DP, waves, p, H, splits, cus))
            return vqi::part_persist(p, H, cus, metric, s);
    }
    if (pair_selected(DP, p.Q, p.tiles_per_split)) return vqi::part_pair(p, H, splits, metric, s);  // single stage: wave pairs
    switch (DP) {
        case 32: return vqi::part_search<32>(waves, p, H, splits, metric, s);
        case 64: return vqi::part_search<64>(waves, p, H, splits, metric, s);
        case 128: return vqi::part_search<128>(waves, p, H, splits, metric, s);
        case 256: return vqi::part_search<256>(waves, p, H, splits, metric, s);
        case 512: return vqi::part_search<512>(waves, p, H, splits, metric, s);
    }
    return fail(VQ_E_UNSUPPORTED, "vq_search: unsupported padded dim");
}


int launch_aux(int DP, const AuxParams &p, int H, int metric, int mode, hipStream_t s) {
    switch (DP) {
        case 32: return vqi::part_aux<32>(p, H, metric, mode, s);
        case 64: return vqi::part_aux<64>(p, H, metric, mode, s);
        case 128: return vqi::part_aux<128>(p, H, metric, mode, s);
        case 256: return vqi::part_aux<256>(p, H, metric, mode, s);
        case 512: return vqi::part_aux<512>(p, H, metric, mode, s);
    }
    return fail(VQ_E_UNSUPPORTED, "vq_sweep_aux: unsupported padded dim");
}


// one slice of a wide-row sweep, by the padded width of the slice
int launch_wide(int wide, int DP, const SearchParams &p, int H, int splits, int metric, hipStream_t s) {
    switch (DP) {
        case 32: return vqi::part_wide<32>(wide, p, H, splits, metric, s);
        case 64: return vqi::part_wide<64>(wide, p, H, splits, metric, s);
        case 128: return vqi::part_wide<128>(wide, p, H, splits, metric, s);
        case 256: return vqi::part_wide<256>(wide, p, H, splits, metric, s);
        case 512: return vqi::part_wide<512>(wide, p, H, splits, metric, s);
    }
    return fail(VQ_E_UNSUPPORTED, "vq_search: unsupported padded dim");
}

int check_common(const vq_args *a) {
    if (!a) return fail(VQ_E_BADARG, "vq: null args");
    if (a->H <= 0 || a->Q <= 0 || a->M < 0 || a->K <= 0 || a->D <= 0) return fail(VQ_E_BADARG, "vq: non-positive size");
    if (a->metric != VQ_METRIC_EUCLID && a->metric != VQ_METRIC_DOT) return fail(VQ_E_BADARG, "vq: unknown metric");
    if (!a->x && a->M > 0) return fail(VQ_E_BADARG, "vq: x is null");
    return 0;
}

// Workspace layout: [keys: H*M int64][loss partials: floats]
// one plane of keys + room for the planes of a K split (few rows: <= 512 x 256 extra keys; a planned split of a mid-size row
// count: a few planes) -- when the planes do not fit, the splits fall back to ONE plane combined with atomic MIN
constexpr long long kKeyPlanesExtraBytes = 8ll << 20;
long long ws_keys_bytes(int H, long long M) { return ((long long)H * M * 8 + 255) / 256 * 256 + kKeyPlanesExtraBytes; }
// residual stacks of many rows: room at the END of the workspace for the residual rows of a tail run stage by stage (plan_residual_tail)
constexpr long long kResidualTailBytes = 64ll << 20;
inline long long residual_tail_room(int H, long long M, int Q) {  // (the row width is not known here: 512 dims, capped)
    if (Q < 2) return 0;
    const long long all = ((long long)H * M * 512 * 4 + 255) / 256 * 256;
    return all < kResidualTailBytes ? all : kResidualTailBytes;
}
// loss partials: one float per wave, stage and 32 rows (16 rows for the wave-pair kernel of 256 < D <= 512)
long long ws_loss_floats(int H, long long M, int Q) { return (long long)H * ((M + 15) / 16 + 16) * Q + (long long)H * 8192 + 64; }

// keys + loss partials of a call, rounded to 256 bytes: what lies in front of the residual rows of a stack's staged tail
long long ws_core_bytes(int H, long long M, int Q) { return (ws_keys_bytes(H, M) + ws_loss_floats(H, M, Q) * 4 + 256 + 255) / 256 * 256; }

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
    p.key_hs = a->M;
    p.pk_bytes = (unsigned)(vq_packed_floats(a->K, a->D) * 4);

    p.ste = (a->flags & VQ_F_STE) ? 1 : 0;
    p.xt = (a->flags & VQ_F_X_F16) ? 1 : ((a->flags & VQ_F_X_BF16) ? 2 : 0);
    p.vec_x = (a->D % 4 == 0 && a->x_rs % 4 == 0 && a->x_hs % 4 == 0 &&
               (p.xt ? (((uintptr_t)a->x & 7) == 0) : aligned16(a->x))) ? 1 : 0;
    p.vec_fin = (p.vec_x && (!a->out || (a->out_rs % 4 == 0 && a->out_hs % 4 == 0 && aligned16(a->out))) &&
                 (!a->cb || (a->cb_hs % 4 == 0 && a->cb_qs % 4 == 0 && aligned16(a->cb)))) ? 1 : 0;
}

// one stage of a residual stack run stage by stage (residual_tail_staged): where the next residual goes, whether `out` accumulates
struct ResidualStage {
    float *res_next;
    int out_acc;
};

int run_finalize(const vq_args *a, const long long *keys, float *loss_part, hipStream_t s, int *nparts_out, int key_planes = 1,
                 const ResidualStage *rst = nullptr) {
    FinalizeParams f;
    memset(&f, 0, sizeof(f));
    if (rst) {
        f.residual = 1;
        f.res_next = rst->res_next;
        f.out_acc = rst->out_acc;
    }
    f.keys = keys;
    f.nparts = key_planes;
    f.part_stride = (long long)a->H * a->M;
    f.x = a->x; f.x_rs = a->x_rs; f.x_hs = a->x_hs;
    f.cb = a->cb; f.cb_hs = a->cb_hs;
    f.out = a->out; f.out_rs = a->out_rs; f.out_hs = a->out_hs;
    f.idx = (long long *)a->idx; f.idx_rs = a->idx_rs; f.idx_hs = a->idx_hs;
    f.best = a->best;
    f.loss_part = loss_part;
    f.M = a->M; f.D = a->D; f.metric = a->metric; f.ste = (a->flags & VQ_F_STE) ? 1 : 0;
    f.vec = (a->D % 4 == 0 && a->cb_hs % 4 == 0 && aligned16(a->cb) && (!a->out || (a->out_rs % 4 == 0 && a->out_hs % 4 == 0 && aligned16(a->out))) &&
             (!(f.ste || loss_part || f.res_next) || (a->x_rs % 4 == 0 && a->x_hs % 4 == 0 && aligned16(a->x))) &&
             (!f.res_next || aligned16(f.res_next))) ? 1 : 0;
    long long blocks = (a->M + 3) / 4;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    if (rst) hipLaunchKernelGGL(vq_finalize_kernel<true>, dim3((unsigned)blocks, (unsigned)a->H), dim3(256), 0, s, f);
    else hipLaunchKernelGGL(vq_finalize_kernel<false>, dim3((unsigned)blocks, (unsigned)a->H), dim3(256), 0, s, f);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "vq_finalize launch");
    if (nparts_out) *nparts_out = (int)(blocks * 4 * a->H);
    return 0;
}

// Rows per workgroup of the fused single-stage launch: 256 (8 waves x 32), 128 (4 wave pairs / 4 waves) at Dp = 512.
// (At Dp <= 128 two 8-wave workgroups share a CU, but a lone one already runs at ~0.93 of the CU's MFMA rate, so a CU is
//  still one slot of the rounds model below, to within a few percent.)
inline int fused_rows_per_wg(int DP) { return DP == 512 ? 128 : 256; }

// Time of one 32-code sub-tile of an 8-wave workgroup, microseconds, from the device's own clock instead of a constant
// measured on one box: two waves share a SIMD and each issues Dp / 2 + 1 fp32 MFMAs of 64 cycles per sub-tile (Dp = 256 at
// 2.4 GHz: 6.9 us; the 7.2 us measured in round 2 included the sweep's ~4 % of vector work).  Only the RATIO between a
// sub-tile and the finalize's HBM time enters the plans.
inline double sub_tile_us(int DP) {
    const DevInfo &di = dev_info();
    const double mhz = di.ok && di.clock_mhz > 0 ? (double)di.clock_mhz : 2400.0;
    return 1.045 * 2.0 * (DP / 2 + 1) * 64.0 / mhz;
}

// Quantisation of the grid: with one workgroup per CU the launch runs in rounds of `cus` workgroups, and a row count just
// above a multiple of cus x rows-per-workgroup pays a whole extra round (M = 70 000 at D = 256: 274 workgroups = 2 rounds for
// 1.07 rounds of work).  Splitting K over S workgroups per row block makes the rounds shorter and fuller at the price of one
// more prologue per split and the keys + finalize tail.  Costs in units of one sub-tile of sweep (8 waves): prologue ~1.5,
// fused finalize ~1.2, keys-init + finalize kernels ~ 3 + rows x D x 8 bytes at ~5 TB/s.  Returns the best S (1 = stay fused).
int plan_k_split(int DP, int H, long long M, int K, int D, int cus, double *cost = nullptr, double *fused_cost = nullptr) {
    const int rpw = fused_rows_per_wg(DP);
    const long long nblk = (M + rpw - 1) / rpw * H;
    const int nsub = (K + kTileCodes - 1) / kTileCodes;
    if (cost) *cost = *fused_cost = (double)((nblk + cus - 1) / cus) * (1.5 + nsub + 1.2);
    if (nblk * 2 <= cus || nsub < 16) return 1;  // (few workgroups: the older rule below splits until the chip is full)
    const double sub_us = sub_tile_us(DP);  // one sub-tile of all the workgroup's waves, microseconds
    const double tail = 3.0 + (double)M * H * D * 8.0 / 5e6 / sub_us;  // keys init + finalize kernels
    auto rounds = [&](long long wgs) { return (double)((wgs + cus - 1) / cus); };
    const double fused = rounds(nblk) * (1.5 + nsub + 1.2);
    double best = fused;
    int best_s = 1;
    for (int S = 2; S <= 16 && S * 8 <= nsub; ++S) {
        const int per = (nsub + S - 1) / S;
        const double t = rounds(nblk * S) * (1.5 + per) + tail;
        if (t < best) {
            best = t;
            best_s = S;
        }
    }
    if (best >= 0.88 * fused) return 1;
    if (cost) *cost = best;
    return best_s;
}

// Third remedy for the same quantisation: the row blocks that fill whole rounds run fused, the
// remainder (fewer blocks than CUs) is searched by a second call, which splits K until the chip is full -- a short round
// instead of a whole one.  Returns the rows of the fused part, 0 when the model does not predict >= 5 % over both alternatives.
long long plan_main_tail(int DP, int H, long long M, int K, int D, int cus) {
    const int rpw = fused_rows_per_wg(DP);
    const long long nblk_h = (M + rpw - 1) / rpw;  // row blocks per head; every head gets the same cut
    long long full = nblk_h * H / cus;
    while (full > 0 && (full * cus) % H) --full;   // whole rounds that are also whole row blocks of every head
    const long long rem = nblk_h * H - full * cus;
    const int nsub = (K + kTileCodes - 1) / kTileCodes;
    if (full < 1 || rem == 0 || rem >= cus || nsub < 8) return 0;
    double plan_cost = 0.0, fused_cost = 0.0;
    plan_k_split(DP, H, M, K, D, cus, &plan_cost, &fused_cost);
    int st = (int)((cus + rem - 1) / rem);
    if (st > nsub / 8) st = nsub / 8;
    if (st < 1) st = 1;
    const int per = (nsub + st - 1) / st;
    const double sub_us = sub_tile_us(DP);
    const double tail = (double)((rem * st + cus - 1) / cus) * (1.5 + per) + 3.0 + (double)(rem * rpw) * D * 8.0 / 5e6 / sub_us;
    // (rem counts workgroups of all heads together)
    const double hybrid = (double)full * (1.5 + nsub + 1.2) + tail;
    const double other = plan_cost < fused_cost ? plan_cost : fused_cost;
    return hybrid < 0.95 * other ? full * cus / H * rpw : 0;
}

// ---- rows wider than 512 dims -------------------------------------------------------------------
// The distance of a (row, code) pair is ONE k-ordered fmaf chain over all dims, so the sweep is cut along d into slices of
// kWideSlice dims: slice j continues the chains slice j - 1 left in the workspace (the accumulators' own fragment layout: every
// lane reads back exactly the 16-byte pieces it wrote, coalesced), and the last slice closes them with the norms and runs
// the argmin into packed keys.  The workspace holds the chains of one (row chunk) x (code chunk) at a time.

struct WidePlan {
    int nd;            // slices
    int kc;            // codes per chunk (multiple of 32)
    long long mc;      // rows per chunk (multiple of kWideRows)
    long long acc_bytes, xn_bytes;
};

WidePlan wide_plan(int H, long long M, int K, int D) {
    WidePlan w;
    w.nd = (D + kWideSlice - 1) / kWideSlice;
    const int Kp = round_up(K, kTileCodes);
    w.kc = Kp < kWideCodes ? Kp : kWideCodes;
    const long long Mp = (M + kWideRows - 1) / kWideRows * kWideRows;
    long long mc = kWideChunkBytes / ((long long)H * w.kc * 4) / kWideRows * kWideRows;
    if (mc < kWideRows) mc = kWideRows;
    if (mc > Mp) mc = Mp;
    w.mc = mc;
    w.acc_bytes = (long long)H * mc * w.kc * 4;
    w.xn_bytes = 2 * (((long long)H * mc * 4 + 255) / 256 * 256);  // two buffers: a slice reads one and writes the other
    return w;
}

long long wide_image_floats(int K) { return (long long)round_up(K, kTileCodes) * (kWideSlice + 4) + kPackSlack; }
// the last slice is padded like a narrow row of its own width (32 ... kWideSlice dims) and packed in that layout
int wide_last_dims(int D) { return D - (D - 1) / kWideSlice * kWideSlice; }
long long wide_last_image_floats(int K, int D) {
    const int DP = padded_dim(wide_last_dims(D));
    return (long long)round_up(K, kTileCodes * sub_tiles(DP)) * (DP + 4) + kPackSlack;
}

bool wide_workspace_ok(const vq_args *a) {
    const WidePlan w = wide_plan(a->H, a->M, a->K, a->D);
    return a->workspace && a->workspace_bytes >= vq_workspace_bytes(a->H, a->M, 1) + w.acc_bytes + w.xn_bytes;
}

// One (row chunk, code chunk) covers all codes and every launch fills the chip without a K split: the last slice can finish
// the inference call itself (idx, best, out = codebook[idx]) instead of going through keys and the finalize kernel.
bool wide_fusable(const vq_args *a) {
    if (a->Q != 1 || (a->flags & (VQ_F_STE | VQ_F_FORCE_SPLIT | VQ_F_FORCE_SIMPLE)) || a->sq_err || !a->out || !a->idx || !a->cb)
        return false;
    if (a->D % 4 || a->out_rs % 4 || a->out_hs % 4 || !aligned16(a->out) || a->cb_hs % 4 || !aligned16(a->cb)) return false;
    if (round_up(a->K, kTileCodes) > kWideCodes) return false;
    const DevInfo &di = dev_info();
    const int cus = di.ok && di.cus > 0 ? di.cus : 256;
    const WidePlan w = wide_plan(a->H, a->M, a->K, a->D);
    const long long last_rows = a->M % w.mc ? a->M % w.mc : w.mc;  // the smallest row chunk
    if (((last_rows + kWideRows - 1) / kWideRows) * a->H < cus) return false;
    // a row count that leaves the last round of workgroups mostly empty is better served by a K split (keys path)
    return plan_k_split(kWideSlice, a->H, w.mc < a->M ? w.mc : a->M, a->K, kWideSlice, cus) == 1;
}

// `keys` (search: argmin into packed keys), `sims` (the similarity matrix itself) or `fused` (the whole inference call)
int run_wide(const vq_args *a, long long idx_offset, long long *keys, float *sims, long long sims_rs, long long sims_hs,
             hipStream_t s, bool fused = false) {
    if (!a->packed) return fail(VQ_E_BADARG, "vq: packed codebook is null");
    if (a->flags & (VQ_F_X_F16 | VQ_F_X_BF16)) return fail(VQ_E_UNSUPPORTED, "vq: 2-byte rows need D <= 512");
    const long long img = wide_image_floats(a->K);
    if (img * 4 >= (1ll << 31)) return fail(VQ_E_UNSUPPORTED, "vq: packed codebook image >= 2 GiB (shard the codebook)");
    const WidePlan w = wide_plan(a->H, a->M, a->K, a->D);
    const long long base = vq_workspace_bytes(a->H, a->M, 1);
    if (!a->workspace || a->workspace_bytes < base + w.acc_bytes + w.xn_bytes)
        return fail(VQ_E_BADARG, "vq: workspace too small for rows wider than 512 dims (see vq_workspace_bytes_wide)");
    float *acc_ws = (float *)((char *)a->workspace + base);
    float *xn_ws = (float *)((char *)a->workspace + base + w.acc_bytes);
    const DevInfo &di = dev_info();
    const int cus = di.ok && di.cus > 0 ? di.cus : 256;
    const int Kp = round_up(a->K, kTileCodes);
    const int d_last = wide_last_dims(a->D), dp_last = padded_dim(d_last);
    for (long long m0 = 0; m0 < a->M; m0 += w.mc) {
        const long long mrows = (a->M - m0 < w.mc) ? a->M - m0 : w.mc;
        const long long nblk = (mrows + kWideRows - 1) / kWideRows;
        for (int k0 = 0; k0 < Kp; k0 += w.kc) {
            const int kcodes = (a->K - k0 < w.kc) ? a->K - k0 : w.kc;  // real codes of this chunk (> 0: k0 < Kp, K > Kp - 32)
            const int nsub = (kcodes + kTileCodes - 1) / kTileCodes;
            for (int j = 0; j < w.nd; ++j) {
                const bool last = j + 1 == w.nd;
                const int DP = last ? dp_last : kWideSlice;
                const int rs = DP + 4;                                        // packed row stride of this slice's image
                const long long img_j = last ? wide_last_image_floats(a->K, a->D) : img;
                SearchParams p;
                memset(&p, 0, sizeof(p));
                p.x = a->x + m0 * a->x_rs + (long long)j * kWideSlice;
                p.x_rs = a->x_rs; p.x_hs = a->x_hs;
                p.packed = a->packed + (long long)j * img + (long long)k0 * rs;
                p.pk_hs = a->pk_hs;
                p.pk_bytes = (unsigned)((img_j - (long long)k0 * rs) * 4);
                p.M = mrows; p.K = kcodes; p.D = last ? d_last : kWideSlice; p.Q = 1;
                const int tc = kTileCodes * sub_tiles(DP);
                p.ntiles = (kcodes + tc - 1) / tc;
                // K is split over workgroups until the chip is full (one workgroup per CU at Dp = 512 / 8 waves at Dp = 256)
                const long long fill = (long long)cus * ((DP == 512 || (kWideWaves == 8 && DP == 256)) ? 1 : 2);
                int splits = 1;
                if (nblk * a->H < fill) {
                    splits = (int)((fill + nblk * a->H - 1) / (nblk * a->H));
                    if (splits > p.ntiles) splits = p.ntiles;
                } else if (!fused) {
                    splits = plan_k_split(DP, a->H, mrows, kcodes, DP, cus);  // partly filled last round of workgroups
                    if (splits > p.ntiles) splits = p.ntiles;
                }
                p.tiles_per_split = (p.ntiles + splits - 1) / splits;
                splits = (p.ntiles + p.tiles_per_split - 1) / p.tiles_per_split;
                p.mode = fused ? kModeFused : kModeKeys;
                if (fused) {
                    p.cb = a->cb; p.cb_hs = a->cb_hs;
                    p.out = a->out + m0 * a->out_rs; p.out_rs = a->out_rs; p.out_hs = a->out_hs;
                    p.idx = (long long *)a->idx + m0 * a->idx_rs; p.idx_rs = a->idx_rs; p.idx_hs = a->idx_hs;
                    p.best = a->best ? a->best + m0 * a->idx_rs : nullptr;
                    p.fin_D = a->D;
                    splits = 1;
                    p.tiles_per_split = p.ntiles;
                }
                p.keys = keys ? keys + m0 : nullptr;
                p.key_hs = a->M;
                p.idx_offset = idx_offset + k0;
                p.vec_x = (p.D % 4 == 0 && a->x_rs % 4 == 0 && a->x_hs % 4 == 0 && aligned16(a->x)) ? 1 : 0;
                p.acc_ws = acc_ws;
                p.ws_hs = nblk * kWideWaves * (long long)nsub * 256;
                p.ws_nsub = nsub;
                p.acc_in = j > 0;
                p.xn_ws = xn_ws + ((j + 1) & 1) * (w.xn_bytes / 8);
                p.xn_out = xn_ws + (j & 1) * (w.xn_bytes / 8);
                p.xn_hs = w.mc;
                if (sims) {
                    p.sims = sims + m0 * sims_rs + k0;
                    p.sims_rs = sims_rs; p.sims_hs = sims_hs;
                    p.vec_s = (a->K % 4 == 0 && sims_rs % 4 == 0 && sims_hs % 4 == 0 && aligned16(sims)) ? 1 : 0;
                }
                const int rc = launch_wide(!last ? 1 : (sims ? 3 : 2), DP, p, a->H, splits, a->metric, s);
                if (rc) return rc;
            }
        }
    }
    return 0;
}

// How a keys-mode search of `a` is launched on this device: workgroup size and the number of K splits (= key planes when the
// splits store into planes of their own instead of combining with atomic MIN).
struct KeysPlan {
    int DP, waves, splits, tiles_per_split;
    bool mfma;  // false: the one-thread-per-row kernel / the sliced sweep of wide rows (one plane, atomic MIN)
};

KeysPlan plan_keys(const vq_args *a, int planned_splits) {
    KeysPlan k;
    k.DP = padded_dim(a->D);
    k.mfma = k.DP != 0 && !(a->flags & VQ_F_FORCE_SIMPLE);
    k.waves = 8;
    k.splits = 1;
    k.tiles_per_split = 1;
    if (!k.mfma) return k;
    const DevInfo &di = dev_info();
    const int cus = di.ok && di.cus > 0 ? di.cus : 256;
    const int tc = kTileCodes * sub_tiles(k.DP);
    const int ntiles = (a->K + tc - 1) / tc;
    k.waves = (k.DP == 512) ? 4 : 8;
    long long wgs = (long long)a->H * ((a->M + 32 * k.waves - 1) / (32 * k.waves));
    if (k.waves == 8 && wgs < cus && planned_splits == 0) {
        k.waves = 4;
        wgs = (long long)a->H * ((a->M + 127) / 128);
    }
    // K is split until the chip is full: two 4-wave workgroups fit a CU at Dp <= 256, one (LDS) at Dp = 512 -- splitting
    // further only repeats the prologue and, in the wave-pair kernel, the extra pipeline step
    const long long fill = (long long)cus * (k.DP == 512 ? 1 : 2);
    int splits = 1;
    if (planned_splits > 0) {
        splits = planned_splits < ntiles ? planned_splits : ntiles;  // (plan_k_split: full-size workgroups, S splits)
    } else if (wgs < fill) {
        splits = (int)((fill + wgs - 1) / wgs);
        if (splits > ntiles) splits = ntiles;
        if (splits < 1) splits = 1;
    }
    k.tiles_per_split = (ntiles + splits - 1) / splits;
    k.splits = (ntiles + k.tiles_per_split - 1) / k.tiles_per_split;  // every split owns at least one tile
    return k;
}

// `part_stride` > 0: K split z stores its keys into plane z (planes part_stride keys apart, plan_keys(..).splits of them,
// nothing to initialise); 0: all splits combine into ONE plane with atomic MIN (the caller initialised it).
int run_search_keys(const vq_args *a, long long idx_offset, long long *keys, hipStream_t s, int planned_splits = 0,
                    long long part_stride = 0) {
    const KeysPlan kp = plan_keys(a, planned_splits);
    if (!kp.mfma && part_stride > 0) return fail(VQ_E_UNSUPPORTED, "vq: key planes need the MFMA kernel (D <= 512)");
    if (kp.DP == 0 && !(a->flags & VQ_F_FORCE_SIMPLE)) return run_wide(a, idx_offset, keys, nullptr, 0, 0, s);
    if (!kp.mfma) {
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
    p.mode = part_stride > 0 ? kModeKeyParts : kModeKeys;
    p.key_zs = part_stride;
    p.keys = keys;
    p.idx_offset = idx_offset;
    p.Q = 1;
    p.out = nullptr;
    p.loss_part = nullptr;
    p.tiles_per_split = kp.tiles_per_split;
    return launch_search(kp.DP, kp.waves, p, a->H, kp.splits, a->metric, s);
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
    if (DP == 0)  // one image per slice, the last one in the layout of its own (padded) width
        return (int64_t)((D - 1) / kWideSlice) * wide_image_floats(K) + wide_last_image_floats(K, D);
    return (int64_t)round_up(K, kTileCodes * sub_tiles(DP)) * (DP + 4) + kPackSlack;
}

int64_t vq_workspace_bytes_wide(int H, int64_t M, int K, int D) {
    if (H <= 0 || M < 0 || K <= 0 || D <= 0) return 0;
    const int64_t base = vq_workspace_bytes(H, M, 1);
    if (padded_dim(D) != 0 || M == 0) return base;
    const WidePlan w = wide_plan(H, M, K, D);
    return base + w.acc_bytes + w.xn_bytes;
}

int64_t vq_workspace_bytes(int H, int64_t M, int Q) {
    if (H <= 0 || M < 0 || Q <= 0) return 0;
    // (+ residual stacks of many rows: room for the residual rows of a tail run stage by stage, see plan_residual_tail)
    return ws_core_bytes(H, M, Q) + residual_tail_room(H, M, Q);
}

int vq_pack_codebooks_f32(const float *cb, int n_codebooks, int64_t cb_stride, int K, int D, int metric, float *packed,
                          void *stream) {
    if (!cb || !packed || n_codebooks <= 0 || K <= 0 || D <= 0) return fail(VQ_E_BADARG, "vq_pack: bad argument");
    if (metric != VQ_METRIC_EUCLID && metric != VQ_METRIC_DOT) return fail(VQ_E_BADARG, "vq_pack: unknown metric");
    const int DP = padded_dim(D);
    if (!aligned16(packed)) return fail(VQ_E_BADARG, "vq_pack: packed buffer must be 16-byte aligned");
    const long long pk_stride = vq_packed_floats(K, D);
    hipStream_t s = (hipStream_t)stream;
    if (DP == 0) {  // rows wider than 512 dims: one image per slice
        const int nd = (D + kWideSlice - 1) / kWideSlice;
        for (int j = 0; j < nd; ++j) {
            const int dp = (j + 1 == nd) ? padded_dim(wide_last_dims(D)) : kWideSlice;
            const int Kp = round_up(K, kTileCodes * sub_tiles(dp));
            hipLaunchKernelGGL(vq_pack_kernel, dim3(Kp / 64 + 1, n_codebooks), dim3(64), 0, s, cb, (long long)cb_stride, K, Kp,
                               D, j * kWideSlice, dp, metric, packed + (long long)j * wide_image_floats(K), pk_stride);
        }
        hipLaunchKernelGGL(vq_pack_flag_kernel, dim3(n_codebooks), dim3(256), 0, s, packed, pk_stride, K, kWideSlice + 4, kWideSlice,
                           nd, wide_image_floats(K), wide_last_image_floats(K, D));
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return hip_fail(e, "vq_pack launch");
        return 0;
    }
    const int Kp = round_up(K, kTileCodes * sub_tiles(DP));
    // grid covers Kp rows plus at least one extra block whose threads zero the over-copy slack
    hipLaunchKernelGGL(vq_pack_kernel, dim3(Kp / 64 + 1, n_codebooks), dim3(64), 0, s, cb, (long long)cb_stride, K, Kp,
                       D, 0, DP, metric, packed, pk_stride);
    hipLaunchKernelGGL(vq_pack_flag_kernel, dim3(n_codebooks), dim3(256), 0, s, packed, pk_stride, K, DP + 4, DP, 1, pk_stride,
                       (long long)vq_packed_floats(K, D));
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

int vq_key_planes(const vq_args *a) {
    if (check_common(a) || a->Q != 1 || a->M == 0) return 1;
    const KeysPlan kp = plan_keys(a, 0);
    return kp.mfma ? kp.splits : 1;
}

int vq_search_key_planes_f32(const vq_args *a, int64_t idx_offset, int64_t *keys, void *stream) {
    int rc = check_common(a);
    if (rc) return rc;
    if (a->Q != 1) return fail(VQ_E_BADARG, "vq_search_key_planes: Q must be 1");
    if (!keys) return fail(VQ_E_BADARG, "vq_search_key_planes: keys is null");
    if (idx_offset < 0 || idx_offset + a->K > 0xFFFFFFFFll) return fail(VQ_E_BADARG, "vq_search_key_planes: index range");
    if (a->M == 0) return 0;
    const KeysPlan kp = plan_keys(a, 0);
    if (!kp.mfma) {  // scalar kernel / wide rows: one plane, combined with atomic MIN
        rc = vq_keys_init(keys, (int64_t)a->H * a->M, stream);
        if (rc) return rc;
        return run_search_keys(a, idx_offset, (long long *)keys, (hipStream_t)stream);
    }
    return run_search_keys(a, idx_offset, (long long *)keys, (hipStream_t)stream, 0, (long long)a->H * a->M);
}

int vq_finalize_key_planes_f32(const vq_args *a, const int64_t *keys, int n_planes, void *stream);

int vq_finalize_keys_f32(const vq_args *a, const int64_t *keys, void *stream) { return vq_finalize_key_planes_f32(a, keys, 1, stream); }

int vq_finalize_key_planes_f32(const vq_args *a, const int64_t *keys, int n_planes, void *stream) {
    int rc = check_common(a);
    if (rc) return rc;
    if (n_planes < 1) return fail(VQ_E_BADARG, "vq_finalize_keys: n_planes must be >= 1");
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
    rc = run_finalize(a, (const long long *)keys, loss_part, s, &nparts, n_planes);
    if (rc) return rc;
    if (a->sq_err) {
        hipLaunchKernelGGL(vq_loss_reduce_kernel, dim3(1), dim3(256), 0, s, loss_part, (long long)nparts, 1, a->sq_err);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return hip_fail(e, "vq_loss_reduce launch");
    }
    return 0;
}

constexpr uint32_t kFlagAccumulateSqErr = 0x80000000u;  // never set by callers: vq_quantize_f32 masks it off

// A single stage as  search into key planes (K split over workgroups: every split stores its winners into a plane of its own, no
// init launch, no atomics; one plane + atomic MIN if the planes do not fit the workspace)  +  finalize (MIN over the planes,
// gather, straight-through, squared error).  `rst`: the stage belongs to a residual stack run stage by stage.
// `sq_err_hs` > 0: one squared-error sum per head, the heads `sq_err_hs` doubles apart (a stage of a grouped stack).
static int split_stage(const vq_args *a, int planned_splits, int acc, void *stream, const ResidualStage *rst, int sq_err_hs = 0) {
    hipStream_t s = (hipStream_t)stream;
    long long *keys = (long long *)a->workspace;
    float *loss_part = (float *)((char *)a->workspace + ws_keys_bytes(a->H, a->M));
    const KeysPlan kp = plan_keys(a, planned_splits);
    int planes = 1;
    if (kp.mfma && (long long)kp.splits * a->H * a->M * 8 <= ws_keys_bytes(a->H, a->M)) planes = kp.splits;
    int rc;
    if (!kp.mfma || planes != kp.splits) {
        rc = vq_keys_init((int64_t *)keys, (int64_t)a->H * a->M, stream);
        if (rc) return rc;
        planes = 1;
    }
    rc = run_search_keys(a, 0, keys, s, planned_splits, kp.mfma && planes == kp.splits ? (long long)a->H * a->M : 0);
    if (rc) return rc;
    int nparts = 0;
    rc = run_finalize(a, keys, a->sq_err ? loss_part : nullptr, s, &nparts, planes, rst);
    if (rc) return rc;
    if (a->sq_err) {
        if (sq_err_hs > 0)  // the finalize's partials are [head][nparts / H]
            hipLaunchKernelGGL(vq_loss_reduce_kernel, dim3(1, a->H), dim3(256), 0, s, loss_part, (long long)(nparts / a->H), 1, a->sq_err, acc, sq_err_hs);
        else
            hipLaunchKernelGGL(vq_loss_reduce_kernel, dim3(1), dim3(256), 0, s, loss_part, (long long)nparts, 1, a->sq_err, acc);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return hip_fail(e, "vq_loss_reduce launch");
    }
    return 0;
}

// ---- residual stacks whose row count leaves the last round of workgroups mostly empty -------------------------------------
// A residual launch cannot split K (every stage needs the whole codebook per row), so M = 70 000 at cfg4's shape pays half a
// round of 128-row workgroups for 7 % of a round of work.  The rows that fill whole rounds run on the fused kernel; the
// remainder runs STAGE BY STAGE, each stage a K-split search over all CUs + a finalize that also writes the next residual
// (r - quant, the fused kernel's arithmetic) into the workspace and accumulates `out`: 2-3 short launches per stage instead of
// half a round of sweep per stage.  The same holds for a stack of FEW rows (less than one round: M = 8192 occupies 32 CUs, or 64
// at one wave per SIMD): then every row runs stage by stage.  Returns the rows (per head) of the fused part -- 0: all rows
// staged -- or -1 = keep the single fused launch.

static long long plan_residual_tail(const vq_args *a, int DP, int cus) {
    if (a->Q < 2 || DP == 0 || (a->flags & (VQ_F_FORCE_SIMPLE | VQ_F_FORCE_SPLIT))) return -1;
    if (getenv("VQ_NO_RESIDUAL_TAIL") != nullptr) return -1;  // (read per call: tests run both plans in one process)
    const int rpw = fused_rows_per_wg(DP);  // 256 (128 at Dp = 512)
    const long long nblk_h = (a->M + rpw - 1) / rpw;
    long long full = nblk_h * a->H / cus;
    while (full > 0 && (full * cus) % a->H) --full;  // whole rounds that are whole row blocks of every head
    const long long rem = nblk_h * a->H - full * cus;  // workgroups of the last, partly filled round
    if (rem == 0) return -1;
    const long long m1 = full * cus / a->H * rpw, mt = a->M - m1;
    if (mt <= 0 || (long long)a->H * mt * a->D * 4 > residual_tail_room(a->H, a->M, a->Q)) return -1;  // (no room for the residual rows)
    const int nsub = (a->K + kTileCodes - 1) / kTileCodes;
    if (nsub < 8) return -1;  // (too short to split)
    const double sweep_us = nsub * sub_tile_us(DP);  // one stage of one round
    // the fused alternative: a whole round, or ~0.55 of one when 128-row workgroups fit one per CU (Dp <= 256: a lone 4-wave
    // workgroup has the matrix pipe to itself, see quantize_impl)
    const long long rem4 = ((mt + 127) / 128) * a->H;
    const double fused_rounds = (DP <= 256 && rem4 <= cus) ? 0.55 : 1.0;
    const double fused_us = fused_rounds * a->Q * sweep_us;
    // staged: per stage ~20 us of launches + finalize, the tail's share of a round of sweep (K split: ~1.4 x for the extra prologues)
    const double staged_us = a->Q * (20.0 + 1.4 * sweep_us * (double)rem / cus);
    return staged_us < 0.8 * fused_us ? m1 : -1;
}

static int residual_tail_staged(const vq_args *a, long long m1, void *stream) {
    const long long mt = a->M - m1;
    // [H][mt][D], behind the keys and the loss partials of the whole call
    float *R = (float *)((char *)a->workspace + ws_core_bytes(a->H, a->M, a->Q));
    for (int q = 0; q < a->Q; ++q) {
        vq_args t = *a;
        t.M = mt;
        t.Q = 1;
        if (q == 0) {
            t.x = a->x + m1 * a->x_rs;
        } else {
            t.x = R;
            t.x_rs = a->D;
            t.x_hs = mt * a->D;
        }
        t.cb = a->cb + (long long)q * a->cb_qs;
        t.packed = a->packed + (long long)q * a->pk_qs;
        if (a->out) t.out = a->out + m1 * a->out_rs;
        t.idx = a->idx + m1 * a->idx_rs + (long long)q * a->idx_qs;
        if (a->best) t.best = a->best + m1 * a->idx_rs + (long long)q * a->idx_qs;
        if (a->sq_err) t.sq_err = a->sq_err + q;
        t.workspace_bytes = ws_core_bytes(a->H, a->M, a->Q);
        ResidualStage rst;
        rst.res_next = (q + 1 < a->Q) ? R : nullptr;
        rst.out_acc = q > 0;
        t.flags &= ~VQ_F_SQERR_PER_HEAD;
        const int by_head_hs = ((a->flags & VQ_F_SQERR_PER_HEAD) && a->sq_err) ? a->Q : 0;  // sq_err is [H][Q]: stage q of head h at h * Q + q
        const int rc = split_stage(&t, 0, /*acc=*/1, stream, &rst, by_head_hs);  // (the fused part wrote sq_err[q]; this adds the tail's sum)
        if (rc) return rc;
    }
    return 0;
}

static int quantize_impl(const vq_args *a, void *stream, float *lse) {
    int rc = check_common(a);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    if (lse && (a->Q != 1 || padded_dim(a->D) == 0 || (a->flags & (VQ_F_FORCE_SIMPLE | VQ_F_FORCE_SPLIT))))
        return fail(VQ_E_UNSUPPORTED, "vq_quantize_lse: needs Q == 1, D <= 512 and the fused MFMA path");
    if ((a->flags & (VQ_F_X_F16 | VQ_F_X_BF16)) &&
        (a->Q != 1 || padded_dim(a->D) == 0 || lse || a->sq_err || (a->flags & (VQ_F_STE | VQ_F_FORCE_SIMPLE))))
        return fail(VQ_E_UNSUPPORTED, "vq_quantize: 2-byte rows are inference only (Q == 1, D <= 512, no STE / sq_err / lse)");
    if (a->M > 0 && !a->idx) return fail(VQ_E_BADARG, "vq_quantize: idx is null");
    if (a->M > 0 && !a->cb) return fail(VQ_E_BADARG, "vq_quantize: natural codebook is null");
    if (a->M == 0) {
        if (a->sq_err) hipMemsetAsync(a->sq_err, 0, sizeof(double) * a->Q * ((a->flags & VQ_F_SQERR_PER_HEAD) ? a->H : 1), s);
        return 0;
    }
    if (!a->workspace || a->workspace_bytes < vq_workspace_bytes(a->H, a->M, a->Q))
        return fail(VQ_E_BADARG, "vq_quantize: workspace too small (see vq_workspace_bytes)");
    long long *keys = (long long *)a->workspace;
    float *loss_part = (float *)((char *)a->workspace + ws_keys_bytes(a->H, a->M));

    const int DP = padded_dim(a->D);
    const bool simple = (a->flags & VQ_F_FORCE_SIMPLE) || DP == 0;
    const DevInfo &di = dev_info();
    const int cus = di.ok && di.cus > 0 ? di.cus : 256;

    const int acc = (a->flags & kFlagAccumulateSqErr) ? 1 : 0;  // (internal: second call of a two-call plan adds its sum)
    const int res_img = (simple || acc) ? 0 : resident_image_for(a, DP, lse != nullptr, cus);
    if (!res_img && !simple && a->Q == 1 && !lse && !acc && !((a->flags & VQ_F_SQERR_PER_HEAD) && a->sq_err) &&
        !(a->flags & (VQ_F_FORCE_SPLIT | VQ_F_X_F16 | VQ_F_X_BF16))) {
        const long long m1 = plan_main_tail(DP, a->H, a->M, a->K, a->D, cus);
        if (m1 > 0 && m1 < a->M) {  // whole rounds fused, then the remainder as its own (K-split) call
            vq_args a1 = *a, a2 = *a;
            a1.M = m1;
            a2.M = a->M - m1;
            a2.x = a->x + m1 * a->x_rs;
            if (a->out) a2.out = a->out + m1 * a->out_rs;
            a2.idx = a->idx + m1 * a->idx_rs;
            if (a->best) a2.best = a->best + m1 * a->idx_rs;
            if (a->sq_err) a2.flags |= kFlagAccumulateSqErr;  // one squared-error sum over both parts, fixed order
            rc = quantize_impl(&a1, stream, nullptr);
            if (rc) return rc;
            return quantize_impl(&a2, stream, nullptr);
        }
    }

    if (!simple && a->Q > 1 && !acc && a->packed && a->workspace_bytes >= vq_workspace_bytes(a->H, a->M, a->Q)) {
        const long long m1 = plan_residual_tail(a, DP, cus);
        if (m1 >= 0 && m1 < a->M) {  // whole rounds on the fused kernel, the remainder stage by stage (K split over all CUs)
            if (m1 > 0) {
                vq_args a1 = *a;
                a1.M = m1;
                rc = quantize_impl(&a1, stream, nullptr);
                if (rc) return rc;
            } else if (a->sq_err) {  // (no fused part: the stages ADD their sums)
                hipError_t e = hipMemsetAsync(a->sq_err, 0, sizeof(double) * a->Q * ((a->flags & VQ_F_SQERR_PER_HEAD) ? a->H : 1), s);
                if (e != hipSuccess) return hip_fail(e, "vq_quantize: clearing sq_err");
            }
            return residual_tail_staged(a, m1, stream);
        }
    }

    // ---- choose fused (one launch, no K split) or split (keys + finalize) ----
    bool fused = !simple;
    int planned_splits = 0;
    int waves = (DP == 512) ? 4 : 8;
    if (fused) {
        long long wgs = (long long)a->H * ((a->M + 32 * waves - 1) / (32 * waves));
        if (waves == 8 && wgs < cus) {
            waves = 4;
            wgs = (long long)a->H * ((a->M + 127) / 128);
        }
        // Residual stacks cannot split K (every stage needs the whole codebook per row), so a row count just above a multiple of
        // cus x 256 used to pay a whole extra round of 8-wave workgroups.  Two 4-wave workgroups share a CU at the pace of one
        // 8-wave workgroup, and a LONE 4-wave workgroup (one wave per SIMD: the matrix pipe to itself) finishes its 128 rows in
        // about half a round -- so with 128-row workgroups the remainder costs half a round instead of a whole one whenever it
        // fits one workgroup per CU (M = 70 000 at cfg4's shape: 2 rounds -> ~1.55).
        if (waves == 8 && DP == 256 && a->Q > 1) {
            const long long nblk4 = (long long)a->H * ((a->M + 127) / 128);
            const long long full = nblk4 / (2ll * cus), rem = nblk4 % (2ll * cus);
            const double t8 = (double)((wgs + cus - 1) / cus);
            const double t4 = (double)full + (rem == 0 ? 0.0 : (rem <= cus ? 0.55 : 1.0));
            if (t4 < 0.97 * t8) {
                waves = 4;
                wgs = nblk4;
            }
        }
#ifdef VQ_EXP_WAVES
        waves = VQ_EXP_WAVES;  // diagnostic builds only
#endif
        const int ntiles = (a->K + kTileCodes * sub_tiles(DP) - 1) / (kTileCodes * sub_tiles(DP));
        // few workgroups and a long sweep: splitting K over workgroups fills the chip (Q == 1 only)
        if (a->Q == 1 && wgs * 2 <= cus && ntiles * sub_tiles(DP) >= 8 && ntiles >= 2) fused = false;
        if ((a->flags & VQ_F_FORCE_SPLIT) && a->Q == 1) fused = false;
        if (fused && a->Q == 1 && !(a->flags & (VQ_F_X_F16 | VQ_F_X_BF16))) {
            planned_splits = plan_k_split(DP, a->H, a->M, a->K, a->D, cus);  // grid quantisation (see plan_k_split)
            if (planned_splits > 1) fused = false;
        }
        if (res_img) fused = true;  // small codebook resident in LDS: 32-row granularity, no plan needed
        if (lse) fused = true;  // the log-sum-exp needs every code of a row in one workgroup
        if ((a->flags & VQ_F_SQERR_PER_HEAD) && a->sq_err) fused = true;  // per-head partial sums exist on this path only
    }

    if (fused) {
        if (!a->packed) return fail(VQ_E_BADARG, "vq_quantize: packed codebook is null");
        if (vq_packed_floats(a->K, a->D) * 4 >= (1ll << 31))
            return fail(VQ_E_UNSUPPORTED, "vq_quantize: packed codebook image >= 2 GiB (shard the codebook)");
        SearchParams p;
        fill_search_params(p, a);
        p.mode = kModeFused;
        p.loss_part = a->sq_err ? loss_part : nullptr;
        p.lse = lse;
        p.res_img_floats = res_img;
        if (res_img) {
            static const char *env = getenv("VQ_RES_SKEW");  // (diagnostic: start-up skew of the second wave per SIMD)
            p.res_skew = env ? atoi(env) : 0;
            const size_t full = ((size_t)res_img + 8 * (DP / 16) * 512) * 4 + 2 * 8 * 32 * 4;  // a slab buffer per slab of a block
            p.res_nbuf = (DP < 128 && full <= 160 * 1024) ? DP / 16 : 1;
        }
        rc = launch_search(DP, waves, p, a->H, 1, a->metric, s);
        if (rc) return rc;
        if (a->sq_err) {
            const bool pair = pair_selected(DP, a->Q, p.tiles_per_split);  // 8 waves (4 pairs) per 128 rows, one partial per wave
            const bool pers = !res_img && persist_selected(DP, waves, p, a->H, 1, cus);  // one partial per wave of the resident workgroups
            const long long rows_per_wg = pair ? 128 : 32ll * waves;
            const long long per_head = pers ? persist_grid_x(a->M, a->H, cus) * 8 : ((a->M + rows_per_wg - 1) / rows_per_wg) * (pair ? 8 : waves);
            const bool by_head = (a->flags & VQ_F_SQERR_PER_HEAD) != 0;
            hipLaunchKernelGGL(vq_loss_reduce_kernel, dim3(a->Q, by_head ? a->H : 1), dim3(256), 0, s, loss_part,
                               by_head ? per_head : per_head * a->H, a->Q, a->sq_err, acc);
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) return hip_fail(e, "vq_loss_reduce launch");
        }
        return 0;
    }

    if (DP == 0 && wide_fusable(a)) return run_wide(a, 0, nullptr, nullptr, 0, 0, s, true);
    if (a->Q != 1) return fail(VQ_E_UNSUPPORTED, "vq_quantize: residual stages need the MFMA kernel (D <= 512)");
    if ((a->flags & VQ_F_SQERR_PER_HEAD) && a->sq_err)
        return fail(VQ_E_UNSUPPORTED, "vq_quantize: per-head squared errors need the MFMA kernel (D <= 512)");
    return split_stage(a, fused ? 0 : planned_splits, acc, stream, nullptr);
}

int vq_quantize_f32(const vq_args *a, void *stream) {
    if (a && (a->flags & kFlagAccumulateSqErr)) {
        vq_args b = *a;
        b.flags &= ~kFlagAccumulateSqErr;
        return quantize_impl(&b, stream, nullptr);
    }
    return quantize_impl(a, stream, nullptr);
}

int vq_quantize_lse_f32(const vq_args *a, float *lse, void *stream) {
    if (!lse && a && a->M > 0) return fail(VQ_E_BADARG, "vq_quantize_lse: lse is null");
    if (a && (a->flags & kFlagAccumulateSqErr)) {
        vq_args b = *a;
        b.flags &= ~kFlagAccumulateSqErr;
        return quantize_impl(&b, stream, lse);
    }
    return quantize_impl(a, stream, lse);
}

int vq_quantize_backward_f32(const vq_args *a, const float *grad_out, int64_t go_rs, int64_t go_hs, const double *grad_sq_err,
                             float *grad_x, int64_t gx_rs, int64_t gx_hs, void *stream) {
    int rc = check_common(a);
    if (rc) return rc;
    if (a->M == 0) return 0;
    if (!a->cb || !a->idx || !grad_x) return fail(VQ_E_BADARG, "vq_quantize_backward: cb / idx / grad_x is null");
    QuantBwdParams p;
    memset(&p, 0, sizeof(p));
    p.x = a->x; p.x_rs = a->x_rs; p.x_hs = a->x_hs;
    p.cb = a->cb; p.cb_hs = a->cb_hs; p.cb_qs = a->cb_qs;
    p.idx = (const long long *)a->idx; p.idx_rs = a->idx_rs; p.idx_hs = a->idx_hs; p.idx_qs = a->idx_qs;
    p.go = grad_out; p.go_rs = go_rs; p.go_hs = go_hs;
    p.g_err = grad_sq_err;
    p.g_err_hs = (a->flags & VQ_F_SQERR_PER_HEAD) ? a->Q : 0;
    p.gx = grad_x; p.gx_rs = gx_rs; p.gx_hs = gx_hs;
    p.M = a->M; p.D = a->D; p.Q = a->Q; p.ste = (a->flags & VQ_F_STE) ? 1 : 0;
    p.vec = (a->D % 4 == 0 && a->x_rs % 4 == 0 && a->x_hs % 4 == 0 && aligned16(a->x) && gx_rs % 4 == 0 && gx_hs % 4 == 0 &&
             aligned16(grad_x) && a->cb_hs % 4 == 0 && a->cb_qs % 4 == 0 && aligned16(a->cb) &&
             (!grad_out || (go_rs % 4 == 0 && go_hs % 4 == 0 && aligned16(grad_out)))) ? 1 : 0;
    long long blocks = (a->M + 3) / 4;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(vq_quantize_backward_kernel, dim3((unsigned)blocks, (unsigned)a->H), dim3(256), 0, (hipStream_t)stream, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "vq_quantize_backward launch");
    return 0;
}

// Owner-computes launch plan: a wave owns cw codes (8 KiB of partial sums) and one of row_blocks contiguous row ranges
struct EmaOwnerPlan {
    int cw;
    long long owners, row_blocks, rows_per_block;
    size_t lds;
};

static bool ema_owner_plan(int H, long long M, int K, int D, EmaOwnerPlan &pl) {
    if (D > 2048) return false;
    const int D4 = (D + 3) & ~3;
    int cw = 2048 / D4;
    if (cw > 64) cw = 64;
    if (cw > K) cw = K;
    pl.cw = cw;
    pl.owners = (K + cw - 1) / cw;
    const DevInfo &di = dev_info();
    const int cus = di.ok && di.cus > 0 ? di.cus : 256;
    long long row_blocks = (16ll * cus) / (pl.owners * H);  // ~16 waves per CU in total
    if (row_blocks < 1) row_blocks = 1;
    long long rows_per_block = (M + row_blocks - 1) / row_blocks;
    if (rows_per_block < 2048) rows_per_block = 2048;
    rows_per_block = (rows_per_block + 63) / 64 * 64;
    pl.rows_per_block = rows_per_block;
    pl.row_blocks = (M + rows_per_block - 1) / rows_per_block;
    const size_t per_wave = (size_t)cw * D4 + ((cw + 3) & ~3) + 64 * 2 + 64;
    pl.lds = per_wave * 4 * 4;
    return true;
}

static int launch_ema_owner(const EmaOwnerPlan &pl, const float *x, int64_t x_rs, int64_t x_hs, const int64_t *idx, int64_t idx_rs,
                            int64_t idx_hs, const uint8_t *mask, int H, int64_t M, int K, int D, float *counts, float *sums,
                            float *part_sums, float *part_counts, hipStream_t s) {
    static thread_local bool attr_done[kMaxDevices] = {};
    if (int arc = allow_big_lds(vq_ema_accumulate_owner_kernel, attr_done)) return arc;
    hipLaunchKernelGGL(vq_ema_accumulate_owner_kernel, dim3((unsigned)((pl.owners + 3) / 4), (unsigned)pl.row_blocks, (unsigned)H),
                       dim3(256), pl.lds, s, x, (long long)x_rs, (long long)x_hs, (const long long *)idx, (long long)idx_rs,
                       (long long)idx_hs, mask, (long long)M, pl.rows_per_block, K, pl.cw, D, counts, sums, part_sums, part_counts);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "vq_ema_accumulate_owner launch");
    return 0;
}

int vq_ema_accumulate_f32(const float *x, int64_t x_rs, int64_t x_hs, const int64_t *idx, int64_t idx_rs, int64_t idx_hs,
                          const uint8_t *mask, int H, int64_t M, int K, int D, float *counts, float *sums, void *stream) {
    if (H <= 0 || M < 0 || K <= 0 || D <= 0 || !counts || !sums) return fail(VQ_E_BADARG, "vq_ema_accumulate: bad argument");
    if (M == 0) return 0;
    if (!x || !idx) return fail(VQ_E_BADARG, "vq_ema_accumulate: null input");
    hipStream_t s = (hipStream_t)stream;
    // Owner-computes path: worth it when each owner sees many rows
    EmaOwnerPlan pl;
    if (ema_owner_plan(H, M, K, D, pl) && M / pl.owners >= 1024)
        return launch_ema_owner(pl, x, x_rs, x_hs, idx, idx_rs, idx_hs, mask, H, M, K, D, counts, sums, nullptr, nullptr, s);
    long long blocks = (M + 3) / 4;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(vq_ema_accumulate_kernel, dim3((unsigned)blocks, (unsigned)H), dim3(256), 0, s, x,
                       (long long)x_rs, (long long)x_hs, (const long long *)idx, (long long)idx_rs, (long long)idx_hs, mask,
                       (long long)M, K, D, counts, sums);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "vq_ema_accumulate launch");
    return 0;
}

int64_t vq_ema_det_workspace_bytes(int H, int64_t M, int K, int D) {
    EmaOwnerPlan pl;
    if (H <= 0 || M <= 0 || K <= 0 || D <= 0 || !ema_owner_plan(H, M, K, D, pl)) return 0;
    return pl.row_blocks * H * (int64_t)K * (D + 1) * 4 + 256;
}

int vq_ema_accumulate_det_f32(const float *x, int64_t x_rs, int64_t x_hs, const int64_t *idx, int64_t idx_rs, int64_t idx_hs,
                              const uint8_t *mask, int H, int64_t M, int K, int D, float *counts, float *sums, void *workspace,
                              int64_t workspace_bytes, void *stream) {
    if (H <= 0 || M < 0 || K <= 0 || D <= 0 || !counts || !sums) return fail(VQ_E_BADARG, "vq_ema_accumulate_det: bad argument");
    if (M == 0) return 0;
    if (!x || !idx) return fail(VQ_E_BADARG, "vq_ema_accumulate_det: null input");
    EmaOwnerPlan pl;
    if (!ema_owner_plan(H, M, K, D, pl)) return fail(VQ_E_UNSUPPORTED, "vq_ema_accumulate_det: D > 2048");
    if (!workspace || workspace_bytes < vq_ema_det_workspace_bytes(H, M, K, D) || ((uintptr_t)workspace & 15))
        return fail(VQ_E_BADARG, "vq_ema_accumulate_det: workspace too small or misaligned (see vq_ema_det_workspace_bytes)");
    hipStream_t s = (hipStream_t)stream;
    float *part_sums = (float *)workspace;
    float *part_counts = part_sums + pl.row_blocks * H * (long long)K * D;
    if (int rc = launch_ema_owner(pl, x, x_rs, x_hs, idx, idx_rs, idx_hs, mask, H, M, K, D, counts, sums, part_sums, part_counts, s))
        return rc;
    const long long n_s = (long long)H * K * D, n_c = (long long)H * K;
    hipLaunchKernelGGL(vq_ema_reduce_parts_kernel, dim3((unsigned)((n_s + 255) / 256)), dim3(256), 0, s, part_sums,
                       (int)pl.row_blocks, n_s, sums);
    hipLaunchKernelGGL(vq_ema_reduce_parts_kernel, dim3((unsigned)((n_c + 255) / 256)), dim3(256), 0, s, part_counts,
                       (int)pl.row_blocks, n_c, counts);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "vq_ema_reduce_parts launch");
    return 0;
}

int vq_ema_accumulate_residual_f32(const vq_args *a, float *counts, float *sums, void *stream) {
    int rc = check_common(a);
    if (rc) return rc;
    if (a->M == 0) return 0;
    if (!a->cb || !a->idx || !counts || !sums) return fail(VQ_E_BADARG, "vq_ema_accumulate_residual: cb / idx / counts / sums is null");
    long long blocks = (a->M + 3) / 4;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(vq_ema_accumulate_residual_kernel, dim3((unsigned)blocks, (unsigned)a->H), dim3(256), 0,
                       (hipStream_t)stream, a->x, (long long)a->x_rs, (long long)a->x_hs, a->cb, (long long)a->cb_hs,
                       (long long)a->cb_qs, (const long long *)a->idx, (long long)a->idx_rs, (long long)a->idx_hs,
                       (long long)a->idx_qs, (long long)a->M, a->K, a->D, a->Q, (a->flags & VQ_F_STE) ? 1 : 0, counts, sums);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "vq_ema_accumulate_residual launch");
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
    // rows wider than 512 dims: the sliced MFMA sweep when the caller provides its workspace (vq_workspace_bytes_wide),
    // else the one-thread-per-entry kernel
    if (DP == 0 && !(a->flags & VQ_F_FORCE_SIMPLE) && a->packed && wide_workspace_ok(a))
        return run_wide(a, 0, nullptr, sims, sims_rs, sims_hs, s);
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
        case 32: return vqi::part_ce_bwd<32>(p, a->H, a->metric, s);
        case 64: return vqi::part_ce_bwd<64>(p, a->H, a->metric, s);
        case 128: return vqi::part_ce_bwd<128>(p, a->H, a->metric, s);
        case 256: return vqi::part_ce_bwd<256>(p, a->H, a->metric, s);
        case 512: return vqi::part_ce_bwd<512>(p, a->H, a->metric, s);
    }
    return fail(VQ_E_UNSUPPORTED, "vq_ce_backward: unsupported padded dim");
}

int vq_max_fused_stages(int D, int want_sq_err) {
    // largest Q one residual launch can hold (winner indices and loss partials of every stage live in LDS); 0: no fused residual launch (D > 512)
    switch (padded_dim(D)) {
        case 32: return max_stages_t<32, 8>(want_sq_err != 0);
        case 64: return max_stages_t<64, 8>(want_sq_err != 0);
        case 128: return max_stages_t<128, 8>(want_sq_err != 0);
        case 256: return max_stages_t<256, 8>(want_sq_err != 0);
        case 512: return max_stages_t<512, 4>(want_sq_err != 0);
    }
    return 0;
}

int vq_nearest_f32(const vq_args *a, void *stream) {
    if (a && a->Q != 1) return fail(VQ_E_BADARG, "vq_nearest_f32: Q must be 1");
    return vq_quantize_f32(a, stream);
}

int vq_residual_f32(const vq_args *a, void *stream) { return vq_quantize_f32(a, stream); }

}  // extern "C"
#endif  // VQ_OWN(0)
