// Evaluation metric of main.py:200-235 on the device: the softmax masks [B,C,h,w] are resized to the annotation's size
// (utils/eval_utils.py:5-12: bilinear, align_corners=True), binarised (> eval_pos_th, or the one-hot of the channel
// argmax when eval_pos_th == -1: main.py:209-219) and compared with the annotation (255 -> 1, 128 -> ignored, else 0:
// main.py:221-224) -- per sample and channel the three integers intersect_and_union needs for the foreground IoU
// (utils/eval_utils.py:14-52,120-123).  One pass over the annotation bytes, no resized mask ever reaches HBM.
// Integer counts with integer atomics: order independent, bit-exact against the reference's numpy histogram.
#include "rcf_common.h"

namespace {

constexpr int CMAX = 8;

__device__ __forceinline__ void src_index_ac(int dst, float scale, int in_size, int &i0, int &i1, float &l1) {
    const float s = scale * dst;                 // align_corners=True (PyTorch area_pixel_compute_source_index)
    i0 = (int)s;
    if (i0 > in_size - 1) i0 = in_size - 1;
    i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
    l1 = s - i0;
}

__global__ void __launch_bounds__(256) eval_counts_kernel(const float *__restrict__ masks, const uint8_t *__restrict__ ann,
                                                          int C, int h, int w, int H, int W, float sh, float sw,
                                                          float pos_th, unsigned long long *__restrict__ counts) {
    __shared__ unsigned sh_cnt[CMAX * 3 + 1];
    const int b = blockIdx.y;
    if (threadIdx.x < CMAX * 3 + 1) sh_cnt[threadIdx.x] = 0u;
    __syncthreads();
    unsigned inter[CMAX], pred[CMAX], lab = 0u;
#pragma unroll
    for (int c = 0; c < CMAX; ++c) { inter[c] = 0u; pred[c] = 0u; }
    const long HW = (long)H * W;
    const float *mb = masks + (long)b * C * h * w;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += (long)gridDim.x * blockDim.x) {
        const unsigned a = ann[(long)b * HW + i];
        if (a == 128u) continue;                                    // ignore_index
        const unsigned l = a == 255u ? 1u : 0u;                     // (ann / 255).long()
        const int y = (int)(i / W), x = (int)(i - (long)y * W);
        int y0, y1, x0, x1;
        float ly, lx;
        src_index_ac(y, sh, h, y0, y1, ly);
        src_index_ac(x, sw, w, x0, x1, lx);
        const float hy = 1.f - ly, hx = 1.f - lx;
        float v[CMAX];
        int best = 0;
#pragma unroll
        for (int c = 0; c < CMAX; ++c) {
            if (c >= C) break;
            const float *p = mb + (long)c * h * w;
            v[c] = hy * (hx * p[y0 * w + x0] + lx * p[y0 * w + x1]) + ly * (hx * p[y1 * w + x0] + lx * p[y1 * w + x1]);
            if (c > 0 && v[c] > v[best]) best = c;                 // torch argmax: first maximum wins
        }
        lab += l;
#pragma unroll
        for (int c = 0; c < CMAX; ++c) {
            if (c >= C) break;
            const unsigned on = pos_th != -1.f ? (v[c] > pos_th ? 1u : 0u) : (c == best ? 1u : 0u);
            pred[c] += on;
            inter[c] += on & l;
        }
    }
    // wavefront sums, then one LDS atomic per wavefront and quantity, then one global atomic per block and quantity
#pragma unroll
    for (int c = 0; c < CMAX; ++c) {
        if (c >= C) break;
        unsigned a = inter[c], p = pred[c];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o); p += __shfl_xor(p, o); }
        if ((threadIdx.x & 63) == 0) { atomicAdd(&sh_cnt[c * 3], a); atomicAdd(&sh_cnt[c * 3 + 1], p); }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) lab += __shfl_xor(lab, o);
    if ((threadIdx.x & 63) == 0) atomicAdd(&sh_cnt[CMAX * 3], lab);
    __syncthreads();
    if (threadIdx.x < C) {
        unsigned long long *o = counts + ((long)b * C + threadIdx.x) * 3;
        atomicAdd(o, (unsigned long long)sh_cnt[threadIdx.x * 3]);
        atomicAdd(o + 1, (unsigned long long)sh_cnt[threadIdx.x * 3 + 1]);
        atomicAdd(o + 2, (unsigned long long)sh_cnt[CMAX * 3]);
    }
}

}  // namespace

/* counts [B][C][3] (uint64, zero-filled by the caller): intersection, prediction area, label area of the foreground
 * class over the non-ignored pixels.  pos_th == -1 exactly (main.py:209 `eval_pos_th != -1`): one-hot of the channel argmax
 * instead of the threshold; any other value, negative ones included, thresholds like the reference. */
extern "C" int rcf_eval_iou_counts_f32(const float *masks, const uint8_t *ann, int B, int C, int h, int w, int H, int W,
                                       float pos_th, unsigned long long *counts, void *stream) {
    if (!masks || !ann || !counts || B <= 0 || C <= 0 || C > CMAX || h <= 0 || w <= 0 || H <= 0 || W <= 0) return RCF_EINVAL;
    const float sh = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f, sw = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
    long blocks = ((long)H * W + 1023) / 1024;
    if (blocks > 256) blocks = 256;
    hipLaunchKernelGGL(eval_counts_kernel, dim3((unsigned)blocks, (unsigned)B), dim3(256), 0, rcf_stream(stream), masks, ann,
                       C, h, w, H, W, sh, sw, pos_th, counts);
    RCF_LAUNCH_CHECK();
    return 0;
}
