/* Plain C against include/vq_mi355x.h: the header must compile as C (no C++-isms), the library must link from C, and the
 * host-side entry points (sizes, limits, argument checks) must work without a GPU.  Built and run by tests/test_cabi_from_c.py. */
#include <stdio.h>
#include <string.h>

#include "vq_mi355x.h"

#define CHECK(cond)                                                     \
    do {                                                                \
        if (!(cond)) {                                                  \
            printf("FAILED line %d: %s\n", __LINE__, #cond);            \
            return 1;                                                   \
        }                                                               \
    } while (0)

int main(void) {
    /* packed image: Kp x (Dp + 4) floats + slack; one image per 512-dim slice beyond 512 dims */
    CHECK(vq_packed_floats(1024, 256) >= 1024 * 260);
    CHECK(vq_packed_floats(1000, 100) >= 1024 * 132);
    CHECK(vq_packed_floats(64, 600) == vq_packed_floats(64, 512) + vq_packed_floats(64, 88));
    {   /* key planes: host arithmetic, one plane when nothing is searched */
        vq_args z;
        memset(&z, 0, sizeof(z));
        CHECK(vq_key_planes(&z) == 1);
        CHECK(vq_finalize_key_planes_f32(&z, NULL, 0, NULL) == VQ_E_BADARG);
    }
    CHECK(vq_packed_floats(0, 32) == 0);
    CHECK(vq_workspace_bytes(1, 1024, 1) >= 1024 * 8);
    CHECK(vq_workspace_bytes(0, 10, 1) == 0);
    CHECK(vq_workspace_bytes_wide(1, 256, 64, 512) == vq_workspace_bytes(1, 256, 1));
    CHECK(vq_workspace_bytes_wide(1, 256, 64, 600) > vq_workspace_bytes(1, 256, 1));
    CHECK(vq_max_fused_stages(256, 0) >= 8);
    CHECK(vq_max_fused_stages(600, 0) == 0);
    /* argument errors are reported without touching a device */
    {
        vq_args a;
        memset(&a, 0, sizeof(a));
        CHECK(vq_quantize_f32(&a, NULL) == VQ_E_BADARG);
        CHECK(strlen(vq_last_error()) > 0);
        a.H = 1; a.Q = 1; a.M = 4; a.K = 8; a.D = 16; a.metric = 7;
        CHECK(vq_quantize_f32(&a, NULL) == VQ_E_BADARG);
        CHECK(strstr(vq_last_error(), "metric") != NULL);
        CHECK(vq_keys_init(NULL, 4, NULL) == VQ_E_BADARG);
    }
    printf("cabi host-only ok\n");
    return 0;
}
