/* A plain C caller of librcf_hip.so: no torch, no C++ -- device memory from hipMalloc, the entry points of include/rcf_hip.h.
 * Built and run by tests/test_abi_gpu.py (hipcc -x c ... -lrcf_hip).  Exit code 0 = every check passed. */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rcf_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d at %s:%d\n", (int)e_, __FILE__, __LINE__); return 2; } } while (0)

/* 4. rcf_conv2d_fwd_f32 on a 4x4 known-answer case (torch.nn.Conv2d, models/resnet.py:164-203): one image, 4 -> 8 channels,
 * 3x3, pad 1, small-integer data, so every product and sum is exact in any fp32 arithmetic and the comparison with the
 * plain C loops below is ==.  Run twice: without operand ranges (bf16-triple kernels) and with ranges from
 * rcf_absmax_f32 plus pre-split weights (fp16-pair kernels, the training step's path).  Then the data gradient
 * (rcf_conv2d_dgrad_f32) and the weight gradient (rcf_conv2d_wgrad_f32) of the same layer against their definitions. */
static int conv_known_answer(void) {
    enum { H = 4, W = 4, CI = 4, CO = 8, R = 3 };
    float hx[H * W * CI], hw[CO * R * R * CI], hb[CO], hy[H * W * CO], ref[H * W * CO], hdx[H * W * CI], rdx[H * W * CI];
    float hdw[CO * R * R * CI], rdw[CO * R * R * CI];
    for (int i = 0; i < H * W * CI; i++) hx[i] = (float)((i * 7 + 3) % 11 - 5);              /* NHWC */
    for (int i = 0; i < CO * R * R * CI; i++) hw[i] = (float)((i * 5 + 1) % 7 - 3);          /* [Cout][R][S][Cin] */
    for (int i = 0; i < CO; i++) hb[i] = (float)(i - 2);
    for (int y = 0; y < H; y++) for (int x = 0; x < W; x++) for (int co = 0; co < CO; co++) {
        float acc = hb[co];
        for (int r = 0; r < R; r++) for (int q = 0; q < R; q++) {
            const int sy = y + r - 1, sx = x + q - 1;
            if (sy < 0 || sy >= H || sx < 0 || sx >= W) continue;
            for (int c = 0; c < CI; c++) acc += hx[(sy * W + sx) * CI + c] * hw[((co * R + r) * R + q) * CI + c];
        }
        ref[(y * W + x) * CO + co] = acc;
    }
    float *dx, *dw, *db, *dy, *ddx, *ddw;
    unsigned *amax;
    void *planes, *ws;
    CK(hipMalloc((void **)&dx, sizeof hx)); CK(hipMalloc((void **)&dw, sizeof hw)); CK(hipMalloc((void **)&db, sizeof hb));
    CK(hipMalloc((void **)&dy, sizeof hy)); CK(hipMalloc((void **)&ddx, sizeof hdx)); CK(hipMalloc((void **)&ddw, sizeof hdw));
    CK(hipMalloc((void **)&amax, 3 * sizeof(unsigned)));
    CK(hipMemcpy(dx, hx, sizeof hx, hipMemcpyHostToDevice)); CK(hipMemcpy(dw, hw, sizeof hw, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, hb, sizeof hb, hipMemcpyHostToDevice));
    rcf_conv_shape s;
    memset(&s, 0, sizeof s);
    s.struct_bytes = (unsigned)sizeof s;
    s.N = 1; s.H = H; s.W = W; s.Cin = CI; s.Ho = H; s.Wo = W; s.Cout = CO; s.R = R; s.S = R; s.stride = 1; s.pad = 1; s.dil = 1;
    s.x_pitch = CI; s.y_pitch = CO;
    const size_t pbytes = rcf_conv_weight_pairs_bytes(CO, CI, R, R);
    CK(hipMalloc(&planes, pbytes));
    size_t wsb = rcf_conv2d_dgrad_workspace_bytes(&s), wsw = rcf_conv2d_wgrad_workspace_bytes(&s);
    if (wsw > wsb) wsb = wsw;
    CK(hipMalloc(&ws, wsb + 256));
    for (int pass = 0; pass < 2; pass++) {
        if (pass == 1) {                         /* operand ranges -> fp16-pair kernels, weights split once */
            CK(hipMemset(amax, 0, 3 * sizeof(unsigned)));
            if (rcf_absmax_f32(dx, H * W, CI, CI, amax, NULL) != 0) return 10;
            if (rcf_absmax_f32(dw, CO * R * R, CI, CI, amax + 1, NULL) != 0) return 10;
            if (rcf_conv_weight_pairs_f32(dw, CO, CI, R, R, amax + 1, planes, 0u, NULL) != 0) return 10;
            s.amax_x = amax; s.amax_w = amax + 1; s.w_pairs = planes;
        }
        CK(hipMemset(dy, 0xff, sizeof hy));
        const int rc = rcf_conv2d_fwd_f32(dx, dw, db, dy, &s, 0, 0.f, 0, NULL);
        if (rc != 0) { printf("rcf_conv2d_fwd_f32 returned %d\n", rc); return 10; }
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(hy, dy, sizeof hy, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int i = 0; i < H * W * CO; i++) bad += hy[i] != ref[i];
        printf("conv2d_fwd 4x4x4->8 3x3 (%s): %d of %d outputs differ from the C loops\n", pass ? "fp16 pairs" : "bf16 triples", bad, H * W * CO);
        if (bad) return 11;
    }
    /* data gradient with dy := ref - bias (integers): dx[sy][sx][c] = sum over taps / co of dy * w */
    for (int i = 0; i < H * W * CO; i++) hy[i] = (float)(((int)ref[i] % 5));
    CK(hipMemcpy(dy, hy, sizeof hy, hipMemcpyHostToDevice));
    memset(rdx, 0, sizeof rdx); memset(rdw, 0, sizeof rdw);
    for (int y = 0; y < H; y++) for (int x = 0; x < W; x++) for (int co = 0; co < CO; co++)
        for (int r = 0; r < R; r++) for (int q = 0; q < R; q++) {
            const int sy = y + r - 1, sx = x + q - 1;
            if (sy < 0 || sy >= H || sx < 0 || sx >= W) continue;
            for (int c = 0; c < CI; c++) {
                rdx[(sy * W + sx) * CI + c] += hy[(y * W + x) * CO + co] * hw[((co * R + r) * R + q) * CI + c];
                rdw[((co * R + r) * R + q) * CI + c] += hy[(y * W + x) * CO + co] * hx[(sy * W + sx) * CI + c];
            }
        }
    CK(hipMemset(amax + 2, 0, sizeof(unsigned)));
    if (rcf_absmax_f32(dy, H * W, CO, CO, amax + 2, NULL) != 0) return 10;
    s.amax_dy = amax + 2;
    if (rcf_conv2d_dgrad_f32(dy, dw, ddx, &s, 0, ws, wsb, NULL) != 0) return 12;
    if (rcf_conv2d_wgrad_f32(dx, dy, ddw, &s, 0, ws, wsb, NULL) != 0) return 12;
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(hdx, ddx, sizeof hdx, hipMemcpyDeviceToHost)); CK(hipMemcpy(hdw, ddw, sizeof hdw, hipMemcpyDeviceToHost));
    int badx = 0, badw = 0;
    for (int i = 0; i < H * W * CI; i++) badx += hdx[i] != rdx[i];
    for (int i = 0; i < CO * R * R * CI; i++) badw += hdw[i] != rdw[i];
    printf("conv2d_dgrad: %d of %d differ; conv2d_wgrad: %d of %d differ\n", badx, H * W * CI, badw, CO * R * R * CI);
    if (badx || badw) return 13;
    /* the identity gradient of a residual join, materialised: out = mask ? g : 0 (bit e of byte [row][c/4] = channel 4 (c/4) + e);
     * here the "join" is the data gradient just computed and the mask takes every other channel */
    {
        unsigned char hm[H * W * CI / 4], *dm;
        float *dout, hout[H * W * CI];
        for (int i = 0; i < H * W * CI / 4; i++) hm[i] = 0x5;
        CK(hipMalloc((void **)&dm, sizeof hm)); CK(hipMalloc((void **)&dout, sizeof hout));
        CK(hipMemcpy(dm, hm, sizeof hm, hipMemcpyHostToDevice));
        if (rcf_relu_mask_copy_mp(ddx, RCF_F32, CI, dm, dout, CI, H * W, CI, 0, NULL) != 0) return 15;
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(hout, dout, sizeof hout, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int i = 0; i < H * W * CI; i++) bad += hout[i] != ((i % 4) % 2 == 0 ? rdx[i] : 0.f);
        printf("relu_mask_copy: %d of %d differ\n", bad, H * W * CI);
        if (bad) return 15;
        /* a 4-channel data gradient has no whole column tile: the fused forms say so and refuse BEFORE launching anything */
        if (rcf_conv2d_dgrad_bnsums_ok(&s) != 0) return 16;
        if (rcf_conv2d_dgrad_add_f32(dy, dw, ddx, &s, 0, dout, CI, dm, NULL, NULL, NULL, 0, NULL) != RCF_EINVAL) return 16;
        hipFree(dm); hipFree(dout);
    }
    /* error codes, not crashes */
    if (rcf_conv2d_fwd_f32(NULL, dw, db, dy, &s, 0, 0.f, 0, NULL) != RCF_EINVAL) return 14;
    s.Cin = 3;
    if (rcf_conv2d_fwd_f32(dx, dw, db, dy, &s, 0, 0.f, 0, NULL) != RCF_EINVAL) return 14;
    hipFree(dx); hipFree(dw); hipFree(db); hipFree(dy); hipFree(ddx); hipFree(ddw); hipFree(amax); hipFree(planes); hipFree(ws);
    return 0;
}

/* 4b. round-5 entry points from plain C: (a) a residual join whose residual is the RAW output of the downsample conv, normalised on the
 * way in (rcf_bn_apply_res_mp; models/resnet.py:293-296) against the same float operations in C loops; (b) a thin classifier conv
 * (1x1, 64 -> 4: csrc/thin.hip behind rcf_conv2d_fwd_f32 / _dgrad_f32 / _wgrad_f32) against C loops. */
static int round5_known_answers(void) {
    enum { ROWS = 24, C = 8 };
    float hx[ROWS * C], hr[ROWS * C], hy[ROWS * C], ref[ROWS * C];
    float mean[C], invstd[C], gamma[C], beta[C], rmean[C], rinvstd[C], rgamma[C], rbeta[C];
    for (int i = 0; i < ROWS * C; i++) { hx[i] = (float)((i * 37) % 23) * 0.25f - 2.5f; hr[i] = (float)((i * 13) % 17) * 0.5f - 4.f; }
    for (int c = 0; c < C; c++) {
        mean[c] = 0.1f * c; invstd[c] = 0.5f + 0.1f * c; gamma[c] = 1.f + 0.05f * c; beta[c] = -0.2f * c;
        rmean[c] = -0.3f * c; rinvstd[c] = 0.25f + 0.05f * c; rgamma[c] = 0.9f; rbeta[c] = 0.1f * c;
    }
    for (int i = 0; i < ROWS * C; i++) {
        const int c = i % C;
        const float o = (hx[i] - mean[c]) * invstd[c] * gamma[c] + beta[c];
        const float r = (hr[i] - rmean[c]) * rinvstd[c] * rgamma[c] + rbeta[c];
        const float v = o + r;
        ref[i] = v > 0.f ? v : 0.f;
    }
    float *dx, *dr, *dy, *dc;
    unsigned char *dm, hm[ROWS * C / 4];
    CK(hipMalloc((void **)&dx, sizeof hx)); CK(hipMalloc((void **)&dr, sizeof hr)); CK(hipMalloc((void **)&dy, sizeof hy));
    CK(hipMalloc((void **)&dc, 8 * C * sizeof(float))); CK(hipMalloc((void **)&dm, sizeof hm));
    CK(hipMemcpy(dx, hx, sizeof hx, hipMemcpyHostToDevice)); CK(hipMemcpy(dr, hr, sizeof hr, hipMemcpyHostToDevice));
    const float *consts[8] = {mean, invstd, gamma, beta, rmean, rinvstd, rgamma, rbeta};
    for (int k = 0; k < 8; k++) CK(hipMemcpy(dc + k * C, consts[k], C * sizeof(float), hipMemcpyHostToDevice));
    rcf_bn_res_norm rn = {dc + 4 * C, dc + 5 * C, dc + 6 * C, dc + 7 * C};
    int rc = rcf_bn_apply_res_mp(dx, RCF_F32, C, dr, C, &rn, dy, RCF_F32, C, ROWS, C, dc, dc + C, dc + 2 * C, dc + 3 * C, 1, NULL, 0, dm, NULL,
                                 NULL, NULL, NULL, 0u, NULL);
    if (rc != 0) { printf("rcf_bn_apply_res_mp returned %d\n", rc); return 30; }
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(hy, dy, sizeof hy, hipMemcpyDeviceToHost)); CK(hipMemcpy(hm, dm, sizeof hm, hipMemcpyDeviceToHost));
    int bad = 0, badm = 0;
    for (int i = 0; i < ROWS * C; i++) {
        bad += !(fabsf(hy[i] - ref[i]) <= 1e-6f * (1.f + fabsf(ref[i])));
        badm += ((hm[i / 4] >> (i % 4)) & 1) != (ref[i] > 0.f);
    }
    printf("bn_apply_res (join normalises its raw residual): %d of %d outputs, %d sign bits differ from the C loops\n", bad, ROWS * C, badm);
    if (bad || badm) return 31;
    /* a residual norm without a residual is an argument error, not a crash */
    if (rcf_bn_apply_res_mp(dx, RCF_F32, C, NULL, 0, &rn, dy, RCF_F32, C, ROWS, C, dc, dc + C, dc + 2 * C, dc + 3 * C, 1, NULL, 0, NULL, NULL,
                            NULL, NULL, NULL, 0u, NULL) != RCF_EINVAL) return 32;
    hipFree(dx); hipFree(dr); hipFree(dy); hipFree(dc); hipFree(dm);
    /* (b) thin conv */
    enum { H = 5, W = 7, CI = 64, CO = 4 };
    static float tx[H * W * CI], tw[CO * CI], tb[CO], ty[H * W * CO], tg[H * W * CO], tdx[H * W * CI], tdw[CO * CI];
    static float ry[H * W * CO], rdx[H * W * CI], rdw[CO * CI];
    for (int i = 0; i < H * W * CI; i++) tx[i] = (float)((i * 29) % 31) / 16.f - 1.f;
    for (int i = 0; i < CO * CI; i++) tw[i] = (float)((i * 11) % 13) / 32.f - 0.2f;
    for (int i = 0; i < CO; i++) tb[i] = 0.5f * i;
    for (int i = 0; i < H * W * CO; i++) tg[i] = (float)((i * 7) % 9) / 8.f - 0.5f;
    for (int p = 0; p < H * W; p++)
        for (int n = 0; n < CO; n++) {
            double a = tb[n];
            for (int c = 0; c < CI; c++) a += (double)tx[p * CI + c] * tw[n * CI + c];
            ry[p * CO + n] = (float)a;
        }
    for (int p = 0; p < H * W; p++)
        for (int c = 0; c < CI; c++) {
            double a = 0;
            for (int n = 0; n < CO; n++) a += (double)tg[p * CO + n] * tw[n * CI + c];
            rdx[p * CI + c] = (float)a;
        }
    for (int n = 0; n < CO; n++)
        for (int c = 0; c < CI; c++) {
            double a = 0;
            for (int p = 0; p < H * W; p++) a += (double)tg[p * CO + n] * tx[p * CI + c];
            rdw[n * CI + c] = (float)a;
        }
    float *gx, *gw, *gb, *gy, *gg, *gdx, *gdw;
    void *ws;
    CK(hipMalloc((void **)&gx, sizeof tx)); CK(hipMalloc((void **)&gw, sizeof tw)); CK(hipMalloc((void **)&gb, sizeof tb));
    CK(hipMalloc((void **)&gy, sizeof ty)); CK(hipMalloc((void **)&gg, sizeof tg)); CK(hipMalloc((void **)&gdx, sizeof tdx));
    CK(hipMalloc((void **)&gdw, sizeof tdw));
    CK(hipMemcpy(gx, tx, sizeof tx, hipMemcpyHostToDevice)); CK(hipMemcpy(gw, tw, sizeof tw, hipMemcpyHostToDevice));
    CK(hipMemcpy(gb, tb, sizeof tb, hipMemcpyHostToDevice)); CK(hipMemcpy(gg, tg, sizeof tg, hipMemcpyHostToDevice));
    rcf_conv_shape s;
    memset(&s, 0, sizeof s);
    s.struct_bytes = sizeof s;
    s.N = 1; s.H = H; s.W = W; s.Cin = CI; s.Ho = H; s.Wo = W; s.Cout = CO; s.R = 1; s.S = 1; s.stride = 1; s.pad = 0; s.dil = 1;
    s.x_pitch = CI; s.y_pitch = CO;
    const size_t need = rcf_conv2d_wgrad_workspace_bytes(&s);
    CK(hipMalloc(&ws, need ? need : 16));
    if (rcf_conv2d_fwd_f32(gx, gw, gb, gy, &s, 0, 0.f, 0, NULL) != 0) return 33;
    if (rcf_conv2d_dgrad_f32(gg, gw, gdx, &s, 0, NULL, 0, NULL) != 0) return 33;
    if (rcf_conv2d_wgrad_f32(gx, gg, gdw, &s, 0, ws, need, NULL) != 0) return 33;
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(ty, gy, sizeof ty, hipMemcpyDeviceToHost)); CK(hipMemcpy(tdx, gdx, sizeof tdx, hipMemcpyDeviceToHost));
    CK(hipMemcpy(tdw, gdw, sizeof tdw, hipMemcpyDeviceToHost));
    int b1 = 0, b2 = 0, b3 = 0;
    for (int i = 0; i < H * W * CO; i++) b1 += !(fabsf(ty[i] - ry[i]) <= 2e-6f * (1.f + fabsf(ry[i])));
    for (int i = 0; i < H * W * CI; i++) b2 += !(fabsf(tdx[i] - rdx[i]) <= 2e-6f * (1.f + fabsf(rdx[i])));
    for (int i = 0; i < CO * CI; i++) b3 += !(fabsf(tdw[i] - rdw[i]) <= 2e-6f * (1.f + fabsf(rdw[i])));
    printf("thin conv 64->4 (workspace %zu B): fwd %d of %d, dgrad %d of %d, wgrad %d of %d differ from the C loops\n", need, b1, H * W * CO,
           b2, H * W * CI, b3, CO * CI);
    if (b1 || b2 || b3) return 34;
    hipFree(gx); hipFree(gw); hipFree(gb); hipFree(gy); hipFree(gg); hipFree(gdx); hipFree(gdw); hipFree(ws);
    return 0;
}

/* 5. rcf_crf_soft (torchcrf_cpp.crf_soft, tools/torchCRF/src/torchcrf.cu:106-126) with caller-owned hipMalloc'd buffers.
 * (a) both potentials' weights 0: no pairwise term, the MAP is the arg-min of the unary energies on every pixel and the
 * marginals are softmax(-U) (torchcrf.cu:28,41: inactive potentials).  (b) the training configuration's potentials
 * (crf_head.py:13-20: weights 3 / 10, sxy 3 / 60, srgb 5 -> scaled for a small frame), 5 iterations, on a two-colour
 * frame whose unary energies already agree with the colour regions: the filter can only reinforce them, MAP unchanged;
 * the lattice vertex counts come back positive.  (c) bad arguments return codes. */
static int crf_known_answer(void) {
    enum { H = 24, W = 40, F = 2 };
    const size_t npx = (size_t)F * H * W;
    unsigned char *hrgb = (unsigned char *)malloc(npx * 3);
    float *hun = (float *)malloc(npx * 2 * sizeof(float)), *hq = (float *)malloc(npx * 2 * sizeof(float));
    short *hmap = (short *)malloc(npx * sizeof(short));
    int hnv[2 * F];
    for (size_t i = 0; i < npx; i++) {
        const int x = (int)(i % W), f = (int)(i / ((size_t)H * W));
        const int fg = f == 0 ? x >= W / 2 : x < W / 3;                  /* the colour regions */
        hrgb[3 * i] = fg ? 200 : 30; hrgb[3 * i + 1] = fg ? 180 : 40; hrgb[3 * i + 2] = fg ? 20 : 90;
        const float conf = 0.6f + 0.3f * (float)((i * 2654435761u) % 97u) / 97.0f;   /* P(label = region) in [0.6, 0.9] */
        hun[2 * i + fg] = -logf(conf);
        hun[2 * i + 1 - fg] = -logf(1.0f - conf);
    }
    unsigned char *drgb; float *dun, *dq; short *dmap; int *dnv; void *ws;
    const size_t wsb = rcf_crf_workspace_bytes(W, H, F);
    CK(hipMalloc((void **)&drgb, npx * 3)); CK(hipMalloc((void **)&dun, npx * 2 * sizeof(float)));
    CK(hipMalloc((void **)&dq, npx * 2 * sizeof(float))); CK(hipMalloc((void **)&dmap, npx * sizeof(short)));
    CK(hipMalloc((void **)&dnv, sizeof hnv)); CK(hipMalloc(&ws, wsb));
    CK(hipMemcpy(drgb, hrgb, npx * 3, hipMemcpyHostToDevice)); CK(hipMemcpy(dun, hun, npx * 2 * sizeof(float), hipMemcpyHostToDevice));
    for (int pass = 0; pass < 2; pass++) {
        const float w_s = pass ? 3.f : 0.f, w_a = pass ? 10.f : 0.f;
        CK(hipMemset(dmap, 0xff, npx * sizeof(short)));
        const int rc = rcf_crf_soft(drgb, dun, W, H, F, w_s, 3.f, w_a, 20.f, 5.f, 5, (int16_t *)dmap, dq, dnv, ws, wsb, NULL);
        if (rc != 0) { printf("rcf_crf_soft returned %d\n", rc); return 20; }
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(hmap, dmap, npx * sizeof(short), hipMemcpyDeviceToHost));
        CK(hipMemcpy(hq, dq, npx * 2 * sizeof(float), hipMemcpyDeviceToHost));
        CK(hipMemcpy(hnv, dnv, sizeof hnv, hipMemcpyDeviceToHost));
        size_t bad = 0; double worst = 0, sum_err = 0;
        for (size_t i = 0; i < npx; i++) {
            const int want = hun[2 * i + 1] < hun[2 * i];
            bad += hmap[i] != want;
            const double s1 = fabs((double)hq[2 * i] + (double)hq[2 * i + 1] - 1.0);
            if (s1 > sum_err) sum_err = s1;
            if (!pass) {                                                  /* softmax(-U) = the confidences themselves */
                const double e0 = exp(-(double)hun[2 * i]), e1 = exp(-(double)hun[2 * i + 1]);
                const double d = fabs((double)hq[2 * i] - e0 / (e0 + e1));
                if (d > worst) worst = d;
            }
        }
        printf("crf_soft %s: %zu of %zu MAP labels differ from the unary arg-min; max |q0+q1-1| = %.2g%s; vertices %d/%d, %d/%d\n",
               pass ? "weights 3 / 10, T=5" : "weights 0", bad, npx, sum_err, pass ? "" : "; q = softmax(-U)", hnv[0], hnv[1], hnv[2], hnv[3]);
        if (bad || !(sum_err < 1e-5) || (!pass && !(worst < 1e-6))) return 21;
        if (pass && (hnv[0] <= 0 || hnv[1] <= 0 || hnv[2] <= 0 || hnv[3] <= 0)) return 22;
        if (!pass && (hnv[0] || hnv[1])) return 22;
    }
    if (rcf_crf_soft(drgb, dun, W, H, F, 3.f, 3.f, 10.f, 20.f, 5.f, 5, (int16_t *)dmap, dq, dnv, ws, wsb / 2, NULL) != RCF_EWORKSPACE) return 23;
    if (rcf_crf_soft(NULL, dun, W, H, F, 3.f, 3.f, 10.f, 20.f, 5.f, 5, (int16_t *)dmap, dq, dnv, ws, wsb, NULL) != RCF_EINVAL) return 23;
    if (rcf_crf_soft_ex(drgb, dun, W, H, F, 3.f, 3.f, 10.f, 20.f, 5.f, 5, 2, (int16_t *)dmap, dq, dnv, ws, wsb, NULL) != RCF_EINVAL) return 23;
    hipFree(drgb); hipFree(dun); hipFree(dq); hipFree(dmap); hipFree(dnv); hipFree(ws);
    free(hrgb); free(hun); free(hq); free(hmap);
    return 0;
}

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
    for (int variant = 0; variant <= 1; variant++) {              /* 0: RCF_WARP_PER_PIXEL, a per-call choice; 1: the tile kernels */
        CK(hipMemset(dy, 0xff, n * sizeof(float)));
        const int rc = rcf_flow_warp_f32(dx, df, dy, B, C, H, W, variant ? 0 : RCF_WARP_PER_PIXEL, NULL);
        if (rc != 0) { printf("rcf_flow_warp_f32 returned %d\n", rc); return 3; }
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(hy, dy, n * sizeof(float), hipMemcpyDeviceToHost));
        double worst = 0;
        for (size_t i = 0; i < n; i++) { const double d = fabs((double)hy[i] - (double)hx[i]); if (d > worst) worst = d; }
        printf("flow_warp variant %d, zero flow: max |out - in| = %.3g\n", variant, worst);
        if (!(worst <= 1e-5)) return 4;   /* fp32 normalise / un-normalise of the grid, as grid_sample */
    }
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
    { const int rc = conv_known_answer(); if (rc) return rc; }
    { const int rc = round5_known_answers(); if (rc) return rc; }
    { const int rc = crf_known_answer(); if (rc) return rc; }
    printf("C ABI smoke: OK\n");
    return 0;
}
