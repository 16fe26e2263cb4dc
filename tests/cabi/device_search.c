/* A consumer of the C ABI that is not Python: plain C + the HIP runtime API.  Packs a codebook, runs vq_quantize_f32 and
 * checks every row against a brute-force double-precision search on the host (the winner must be the true nearest code up
 * to fp32 rounding of the distance, and the quantized row must be that code's row).  Built and run by
 * tests/test_cabi_from_c.py on the GPU box. */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "vq_mi355x.h"

#define HIPCHECK(e)                                                              \
    do {                                                                         \
        hipError_t _e = (e);                                                     \
        if (_e != hipSuccess) {                                                  \
            printf("HIP error %d at line %d\n", (int)_e, __LINE__);              \
            return 2;                                                            \
        }                                                                        \
    } while (0)

static float frand(unsigned *s) {
    *s = *s * 1664525u + 1013904223u;
    return (float)((*s >> 8) & 0xFFFF) / 32768.0f - 1.0f;
}

static int run_case(int M, int K, int D) {
    unsigned seed = 12345u + (unsigned)(M + K + D);
    const size_t nx = (size_t)M * D, nc = (size_t)K * D;
    float *x = (float *)malloc(nx * 4), *cb = (float *)malloc(nc * 4), *out = (float *)malloc(nx * 4);
    int64_t *idx = (int64_t *)malloc((size_t)M * 8);
    for (size_t i = 0; i < nx; ++i) x[i] = frand(&seed);
    for (size_t i = 0; i < nc; ++i) cb[i] = frand(&seed);

    float *dx, *dcb, *dpacked, *dout;
    int64_t *didx;
    void *dws;
    const int64_t pf = vq_packed_floats(K, D), wsb = vq_workspace_bytes_wide(1, M, K, D);
    HIPCHECK(hipMalloc((void **)&dx, nx * 4));
    HIPCHECK(hipMalloc((void **)&dcb, nc * 4));
    HIPCHECK(hipMalloc((void **)&dpacked, (size_t)pf * 4));
    HIPCHECK(hipMalloc((void **)&dout, nx * 4));
    HIPCHECK(hipMalloc((void **)&didx, (size_t)M * 8));
    HIPCHECK(hipMalloc(&dws, (size_t)wsb));
    HIPCHECK(hipMemcpy(dx, x, nx * 4, hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(dcb, cb, nc * 4, hipMemcpyHostToDevice));

    if (vq_pack_codebooks_f32(dcb, 1, (int64_t)nc, K, D, VQ_METRIC_EUCLID, dpacked, NULL) != 0) {
        printf("pack failed: %s\n", vq_last_error());
        return 3;
    }
    vq_args a;
    memset(&a, 0, sizeof(a));
    a.H = 1; a.Q = 1; a.M = M; a.K = K; a.D = D; a.metric = VQ_METRIC_EUCLID;
    a.x = dx; a.x_rs = D; a.x_hs = (int64_t)nx;
    a.cb = dcb; a.cb_hs = (int64_t)nc; a.cb_qs = (int64_t)nc;
    a.packed = dpacked; a.pk_hs = pf; a.pk_qs = pf;
    a.out = dout; a.out_rs = D; a.out_hs = (int64_t)nx;
    a.idx = didx; a.idx_rs = 1; a.idx_hs = M; a.idx_qs = 1;
    a.workspace = dws; a.workspace_bytes = wsb;
    if (vq_quantize_f32(&a, NULL) != 0) {
        printf("quantize failed: %s\n", vq_last_error());
        return 3;
    }
    HIPCHECK(hipDeviceSynchronize());
    HIPCHECK(hipMemcpy(out, dout, nx * 4, hipMemcpyDeviceToHost));
    HIPCHECK(hipMemcpy(idx, didx, (size_t)M * 8, hipMemcpyDeviceToHost));

    int bad = 0;
    for (int m = 0; m < M && bad < 5; ++m) {
        double best = 1e300, got = 0.0;
        for (int k = 0; k < K; ++k) {
            double s = 0.0;
            for (int d = 0; d < D; ++d) {
                const double t = (double)x[(size_t)m * D + d] - (double)cb[(size_t)k * D + d];
                s += t * t;
            }
            if (s < best) best = s;
            if (k == idx[m]) got = s;
        }
        if (idx[m] < 0 || idx[m] >= K || sqrt(got) > sqrt(best) * (1.0 + 1e-5) + 1e-6) {
            printf("row %d: index %lld at distance %.9g, nearest is %.9g\n", m, (long long)idx[m], sqrt(got), sqrt(best));
            ++bad;
        } else if (memcmp(out + (size_t)m * D, cb + (size_t)idx[m] * D, (size_t)D * 4) != 0) {
            printf("row %d: quantized row is not codebook[%lld]\n", m, (long long)idx[m]);
            ++bad;
        }
    }
    hipFree(dx); hipFree(dcb); hipFree(dpacked); hipFree(dout); hipFree(didx); hipFree(dws);
    free(x); free(cb); free(out); free(idx);
    printf("M=%d K=%d D=%d: %s\n", M, K, D, bad ? "MISMATCH" : "ok");
    return bad ? 1 : 0;
}

int main(void) {
    char info[128];
    if (vq_device_info(info, sizeof(info)) != 0) {
        printf("no device: %s\n", vq_last_error());
        return 4;
    }
    printf("%s\n", info);
    if (run_case(1000, 100, 48)) return 1;      /* one launch, fused finalize */
    if (run_case(3000, 1024, 256)) return 1;    /* few row blocks: K split + packed keys + finalize */
    if (run_case(700, 300, 700)) return 1;      /* rows wider than 512 dims: sliced sweep */
    printf("cabi device ok\n");
    return 0;
}
