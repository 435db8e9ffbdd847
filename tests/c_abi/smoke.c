/* A plain C caller of librcf_hip.so: no torch, no C++ -- device memory from hipMalloc, the entry points of include/rcf_hip.h.
 * Built and run by tests/test_abi_gpu.py (hipcc -x c ... -lrcf_hip).  Exit code 0 = every check passed. */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rcf_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d at %s:%d\n", (int)e_, __FILE__, __LINE__); return 2; } } while (0)

int main(void) {
    const int B = 2, C = 3, H = 37, W = 53;
    const size_t n = (size_t)B * C * H * W, nf = (size_t)B * 2 * H * W;
    float *hx = (float *)malloc(n * sizeof(float)), *hy = (float *)malloc(n * sizeof(float)), *hf = (float *)calloc(nf, sizeof(float));
    for (size_t i = 0; i < n; i++) hx[i] = (float)((i * 2654435761u) % 1000u) / 1000.0f;
    float *dx, *dy, *df;
    CK(hipMalloc((void **)&dx, n * sizeof(float)));
    CK(hipMalloc((void **)&dy, n * sizeof(float)));
    CK(hipMalloc((void **)&df, nf * sizeof(float)));
    CK(hipMemcpy(dx, hx, n * sizeof(float), hipMemcpyHostToDevice));
    CK(hipMemcpy(df, hf, nf * sizeof(float), hipMemcpyHostToDevice));
    /* 1. zero flow: the backward warp is the identity (utils/warp_utils.py:84-94), on both kernel variants */
    for (int variant = 0; variant <= 1; variant++) {
        if (rcf_warp_set_variant(variant) != 0) { printf("set_variant failed\n"); return 3; }
        CK(hipMemset(dy, 0xff, n * sizeof(float)));
        const int rc = rcf_flow_warp_f32(dx, df, dy, B, C, H, W, 0, NULL);
        if (rc != 0) { printf("rcf_flow_warp_f32 returned %d\n", rc); return 3; }
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(hy, dy, n * sizeof(float), hipMemcpyDeviceToHost));
        double worst = 0;
        for (size_t i = 0; i < n; i++) { const double d = fabs((double)hy[i] - (double)hx[i]); if (d > worst) worst = d; }
        printf("flow_warp variant %d, zero flow: max |out - in| = %.3g\n", variant, worst);
        if (!(worst <= 1e-5)) return 4;   /* fp32 normalise / un-normalise of the grid, as grid_sample */
    }
    rcf_warp_set_variant(1);
    /* 2. a shift by one pixel to the right: out[y][x] = in[y][x+1] inside the image */
    for (size_t b = 0; b < (size_t)B; b++)
        for (size_t p = 0; p < (size_t)H * W; p++) hf[(b * 2) * H * W + p] = 1.0f;
    CK(hipMemcpy(df, hf, nf * sizeof(float), hipMemcpyHostToDevice));
    if (rcf_flow_warp_f32(dx, df, dy, B, C, H, W, 0, NULL) != 0) return 3;
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(hy, dy, n * sizeof(float), hipMemcpyDeviceToHost));
    double worst = 0;
    for (int b = 0; b < B * C; b++)
        for (int y = 0; y < H; y++)
            for (int x = 0; x + 1 < W; x++) {
                const double d = fabs((double)hy[((size_t)b * H + y) * W + x] - (double)hx[((size_t)b * H + y) * W + x + 1]);
                if (d > worst) worst = d;
            }
    printf("flow_warp, flow = (+1, 0): max |out[x] - in[x+1]| = %.3g\n", worst);
    if (!(worst <= 1e-4)) return 5;   /* the grid is normalised to [-1,1] and back in fp32, as grid_sample does */
    /* 3. argument checking returns codes, never crashes; sizes are plain integers */
    if (rcf_flow_warp_f32(NULL, df, dy, B, C, H, W, 0, NULL) != RCF_EINVAL) return 6;
    if (rcf_flow_warp_f32(dx, df, dy, B, C, H, W, 7, NULL) != RCF_EINVAL) return 6;
    const size_t ws = rcf_crf_workspace_bytes(W, H, B);
    printf("rcf_crf_workspace_bytes(%d, %d, %d) = %zu\n", W, H, B, ws);
    if (ws == 0 || rcf_crf_workspace_bytes(0, H, B) != 0) return 7;
    hipFree(dx); hipFree(dy); hipFree(df);
    free(hx); free(hy); free(hf);
    printf("C ABI smoke: OK\n");
    return 0;
}
