// Training / evaluation data transform of dataset/transforms.py:884-924 (`Transform`) on the device, one fused gather
// per output tensor: Resize (:170-237: mmcv.imrescale = cv2.resize, bilinear for the frames, nearest for the flow /
// pseudo-label fields) -> RandomCrop (:442-508) -> RandomFlip (:249-306) -> PhotoMetricDistortion (:557-687) ->
// NumpyToTensor (/255, CHW: :793-808) -> TorchNormalize (:850-863); FlowTransform (:825-848) and PLTransform (:865-876)
// for the fields.  The random decisions are drawn on the host in the reference's order (rcf_amd/data_pipeline.py) and
// arrive as one rcf_aug_params per sample; the decoded u8 frames never make a round trip through a resized or cropped
// intermediate: every output pixel is computed from its four source taps.
//
// Arithmetic mirrors the CPU operators to the bit, including their 8-bit round trips:
//   * cv2 INTER_LINEAR on 8-bit data: 11-bit fixed-point coefficients, horizontal pass in int, vertical pass
//     ((b0*(r0>>4))>>16 + (b1*(r1>>4))>>16 + 2) >> 2;  INTER_NEAREST: min(floor(dst * src/dst_size), src-1);
//   * convert_one_img: float32(x)*alpha + beta, clip to [0,255], truncate to u8 -- after EVERY photometric stage;
//   * cv2 RGB2HSV on 8-bit data (integer division tables, H in [0,180)), HSV2RGB through fp32, rounded to nearest even;
//   * hue: (int(h) + delta) % 180 in fp64 with numpy's sign rule, truncated to u8;
//   * x/255 and (x - mean)/std as correctly rounded fp32 divisions (the library is built with -ffp-contract=off).
// HBM-bound gather: algorithmic bytes per output pixel = 12 written (3 fp32 planes) + the source bytes it covers
// (3 x the inverse scale squared, about 4.5 at 480->392); the 4-tap byte reads hit L2.
#include "rcf_common.h"

namespace {

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// cv2 resize INTER_LINEAR tap: `horizontal` resets the fraction at the borders, the vertical table only clips the rows
__device__ __forceinline__ void linear_tap(int d, int dst, int src, bool horizontal, int &i0, int &i1, int &a0, int &a1) {
    const double scale = 1.0 / ((double)dst / (double)src);
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= (float)s;
    if (horizontal) {
        if (s < 0) { f = 0.f; s = 0; }
        if (s >= src - 1) { f = 0.f; s = src - 1; }
    }
    a0 = (int)rintf((1.f - f) * 2048.f);
    a1 = (int)rintf(f * 2048.f);
    i0 = clampi(s, 0, src - 1);
    i1 = clampi(s + 1, 0, src - 1);
}

__device__ __forceinline__ int nearest_tap(int d, int dst, int src) {
    const double inv = 1.0 / ((double)dst / (double)src);
    const int s = (int)floor((double)d * inv);
    return s < src - 1 ? s : src - 1;
}

// convert_one_img (dataset/transforms.py:590-594)
__device__ __forceinline__ int convert_u8(int x, float alpha, float beta) {
    float v = (float)x * alpha + beta;
    v = fminf(fmaxf(v, 0.f), 255.f);
    return (int)v;
}

struct Tables {
    int sdiv[256];
    int hdiv[256];
};

__device__ __forceinline__ void rgb2hsv_u8(int r, int g, int b, const int *sdiv, const int *hdiv, int &h, int &s, int &v) {
    v = max(max(r, g), b);
    const int vmin = min(min(r, g), b);
    const int diff = v - vmin;
    s = (diff * sdiv[v] + (1 << 11)) >> 12;
    int hh = v == r ? g - b : (v == g ? b - r + 2 * diff : r - g + 4 * diff);
    hh = (hh * hdiv[diff] + (1 << 11)) >> 12;
    h = hh + (hh < 0 ? 180 : 0);
}

__device__ __forceinline__ int round_u8(float v) {
    const float r = rintf(v * 255.f);
    return (int)fminf(fmaxf(r, 0.f), 255.f);
}

__device__ __forceinline__ void hsv2rgb_u8(int h8, int s8, int v8, int &r, int &g, int &b) {
    const float s = (float)s8 * (1.f / 255.f), v = (float)v8 * (1.f / 255.f);
    if (s8 == 0) { r = g = b = round_u8(v); return; }
    float h = (float)h8 * (6.f / 180.f);
    int sector = (int)floorf(h);
    h -= (float)sector;
    if ((unsigned)sector >= 6u) { sector = 0; h = 0.f; }
    const float t0 = v, t1 = v * (1.f - s), t2 = v * (1.f - s * h), t3 = v * (1.f - s * (1.f - h));
    float fb, fg, fr;
    switch (sector) {
        case 0: fb = t1; fg = t3; fr = t0; break;
        case 1: fb = t1; fg = t0; fr = t2; break;
        case 2: fb = t3; fg = t0; fr = t1; break;
        case 3: fb = t0; fg = t2; fr = t1; break;
        case 4: fb = t0; fg = t1; fr = t3; break;
        default: fb = t2; fg = t1; fr = t0; break;
    }
    r = round_u8(fr); g = round_u8(fg); b = round_u8(fb);
}

struct Norm { float mean[3], std[3]; };

// One thread per output pixel: 64 consecutive x per wavefront, so the three fp32 plane stores are 256-byte coalesced.
__global__ void __launch_bounds__(256) aug_frames_kernel(const uint8_t *__restrict__ src, int B, int I, int H, int W,
                                                         const rcf_aug_params *__restrict__ prm,
                                                         float *__restrict__ out, int oh, int ow, Norm nm) {
    __shared__ Tables tb;
    const int bi = blockIdx.z, b = bi / I, i = bi - b * I;
    const rcf_aug_params p = prm[b];
    const bool photo = (p.ops & 15) != 0;
    if (p.ops & 12) {                                  // saturation / hue need cv2's integer division tables
        for (int t = threadIdx.x; t < 256; t += blockDim.x) {
            tb.sdiv[t] = t ? (int)rint((double)(255 << 12) / (double)t) : 0;
            tb.hdiv[t] = t ? (int)rint((double)(180 << 12) / (6.0 * (double)t)) : 0;
        }
        __syncthreads();
    }
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= ow || y >= oh) return;
    // flip acts on the crop, the crop on the resized frame
    const int rx = p.crop_x + (p.flip ? ow - 1 - x : x), ry = p.crop_y + y;
    int x0, x1, ax0, ax1, y0, y1, ay0, ay1;
    linear_tap(rx, p.rw, W, true, x0, x1, ax0, ax1);
    linear_tap(ry, p.rh, H, false, y0, y1, ay0, ay1);
    const uint8_t *f = src + ((long)b * I + i) * H * W * 3;
    const uint8_t *p00 = f + ((long)y0 * W + x0) * 3, *p01 = f + ((long)y0 * W + x1) * 3;
    const uint8_t *p10 = f + ((long)y1 * W + x0) * 3, *p11 = f + ((long)y1 * W + x1) * 3;
    int c[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int r0 = p00[k] * ax0 + p01[k] * ax1, r1 = p10[k] * ax0 + p11[k] * ax1;
        const int v = (((ay0 * (r0 >> 4)) >> 16) + ((ay1 * (r1 >> 4)) >> 16) + 2) >> 2;
        c[k] = clampi(v, 0, 255);
    }
    if (photo) {
        if (p.ops & 1) {                               // brightness (:599-606)
#pragma unroll
            for (int k = 0; k < 3; ++k) c[k] = convert_u8(c[k], 1.f, p.beta);
        }
        if ((p.ops & 2) && !(p.ops & 16)) {            // contrast first (mode 1, :665-667)
#pragma unroll
            for (int k = 0; k < 3; ++k) c[k] = convert_u8(c[k], p.alpha_c, 0.f);
        }
        if (p.ops & 4) {                               // saturation (:617-632)
            int h, s, v;
            rgb2hsv_u8(c[0], c[1], c[2], tb.sdiv, tb.hdiv, h, s, v);
            s = convert_u8(s, p.alpha_s, 0.f);
            hsv2rgb_u8(h, s, v, c[0], c[1], c[2]);
        }
        if (p.ops & 8) {                               // hue (:634-648)
            int h, s, v;
            rgb2hsv_u8(c[0], c[1], c[2], tb.sdiv, tb.hdiv, h, s, v);
            double hd = fmod((double)h + p.hue_delta, 180.0);
            if (hd != 0.0 && hd < 0.0) hd += 180.0;    // numpy's remainder takes the divisor's sign
            h = (int)hd;                               // 180.0 (a -0.0-sized negative remainder) stays 180, as in numpy
            hsv2rgb_u8(h, s, v, c[0], c[1], c[2]);
        }
        if ((p.ops & 2) && (p.ops & 16)) {             // contrast last (mode 0, :677-679)
#pragma unroll
            for (int k = 0; k < 3; ++k) c[k] = convert_u8(c[k], p.alpha_c, 0.f);
        }
    }
    float *o = out + (((long)i * B + b) * 3) * oh * ow + (long)y * ow + x;
#pragma unroll
    for (int k = 0; k < 3; ++k) o[(long)k * oh * ow] = ((float)c[k] / 255.f - nm.mean[k]) / nm.std[k];
}

// flows [B][K][H][W][2] fp32 -> [K][B][2][oh][ow]; masks [B][K][H][W] u8 -> [K][B][oh][ow] fp32 (/255)
template <bool FLOW>
__global__ void __launch_bounds__(256) aug_fields_kernel(const void *__restrict__ src_, int B, int K, int H, int W,
                                                         const rcf_aug_params *__restrict__ prm,
                                                         float *__restrict__ out, int oh, int ow) {
    const int bk = blockIdx.z, b = bk / K, k = bk - b * K;
    const rcf_aug_params p = prm[b];
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= ow || y >= oh) return;
    const int rx = p.crop_x + (p.flip ? ow - 1 - x : x), ry = p.crop_y + y;
    const long s = (((long)b * K + k) * H + nearest_tap(ry, p.rh, H)) * W + nearest_tap(rx, p.rw, W);
    if (FLOW) {
        const float2 v = ((const float2 *)src_)[s];
        float *o = out + (((long)k * B + b) * 2) * oh * ow + (long)y * ow + x;
        o[0] = v.x * p.flow_sx;
        o[(long)oh * ow] = v.y * p.flow_sy;
    } else {
        out[((long)k * B + b) * oh * ow + (long)y * ow + x] = (float)((const uint8_t *)src_)[s] / 255.f;
    }
}

int check_params_host(int B, int H, int W, int oh, int ow) {
    return (B > 0 && H > 0 && W > 0 && oh > 0 && ow > 0) ? 0 : RCF_EINVAL;
}

}  // namespace

extern "C" int rcf_aug_frames_u8(const uint8_t *frames, int B, int I, int H, int W, const rcf_aug_params *params,
                                 float *out, int oh, int ow, const float *mean3, const float *std3, void *stream) {
    if (!frames || !params || !out || !mean3 || !std3 || I <= 0 || check_params_host(B, H, W, oh, ow)) return RCF_EINVAL;
    if ((long)B * I > 65535) return RCF_EINVAL;
    Norm nm;
    for (int k = 0; k < 3; ++k) { nm.mean[k] = mean3[k]; nm.std[k] = std3[k]; }
    hipLaunchKernelGGL(aug_frames_kernel, dim3(rcf_cdiv(ow, 64), rcf_cdiv(oh, 4), B * I), dim3(256), 0, rcf_stream(stream),
                       frames, B, I, H, W, params, out, oh, ow, nm);
    RCF_LAUNCH_CHECK();
    return 0;
}

extern "C" int rcf_aug_flows_f32(const float *flows, int B, int K, int H, int W, const rcf_aug_params *params, float *out,
                                 int oh, int ow, void *stream) {
    if (!flows || !params || !out || K <= 0 || check_params_host(B, H, W, oh, ow) || (long)B * K > 65535) return RCF_EINVAL;
    hipLaunchKernelGGL(aug_fields_kernel<true>, dim3(rcf_cdiv(ow, 64), rcf_cdiv(oh, 4), B * K), dim3(256), 0,
                       rcf_stream(stream), (const void *)flows, B, K, H, W, params, out, oh, ow);
    RCF_LAUNCH_CHECK();
    return 0;
}

extern "C" int rcf_aug_masks_u8(const uint8_t *masks, int B, int K, int H, int W, const rcf_aug_params *params, float *out,
                                int oh, int ow, void *stream) {
    if (!masks || !params || !out || K <= 0 || check_params_host(B, H, W, oh, ow) || (long)B * K > 65535) return RCF_EINVAL;
    hipLaunchKernelGGL(aug_fields_kernel<false>, dim3(rcf_cdiv(ow, 64), rcf_cdiv(oh, 4), B * K), dim3(256), 0,
                       rcf_stream(stream), (const void *)masks, B, K, H, W, params, out, oh, ow);
    RCF_LAUNCH_CHECK();
    return 0;
}
