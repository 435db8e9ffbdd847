// Fused optimiser / EMA passes over one flat fp32 parameter buffer (HBM-bound, float4 per thread).
// Reference: torch.optim.Adam with coupled weight decay built at main.py:299-307 (lr 1e-4, wd 1e-4,
// betas (0.9,0.999), eps 1e-8, no amsgrad); EMA lerp utils/model_utils.py:33-38 (332 tiny launches in
// the reference, one here).
#include "rcf_common.h"

namespace {
inline int ew_blocks(long total) {
    long b = (total + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

__global__ void __launch_bounds__(256) adam_kernel(float *__restrict__ p, const float *__restrict__ g,
                                                   float *__restrict__ m, float *__restrict__ v, long n, float lr,
                                                   float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt,
                                                   float gscale) {
    const long step = (long)gridDim.x * blockDim.x;
    const float step_size = lr / bc1;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step) {
        float grad = g[i] * gscale;
        const float w = p[i];
        grad = grad + wd * w;                               // coupled L2 (torch Adam weight_decay)
        const float mi = m[i] + (1.f - b1) * (grad - m[i]);  // exp_avg.lerp_(grad, 1-beta1)
        const float vi = b2 * v[i] + (1.f - b2) * grad * grad;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = w - step_size * (mi / denom);
    }
}

__global__ void __launch_bounds__(256) ema_kernel(float *__restrict__ d, const float *__restrict__ s, long n, float m) {
    const long step = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step) d[i] = d[i] * m + s[i] * (1.0f - m);
}

// every entry of a state dict in ONE launch: a workgroup takes one chunk {dst, src, count, kind} of the table (kind 0: fp32,
// d = d m + s (1 - m); kind 1: int64 counters -- num_batches_tracked -- d = trunc(float(d) m + float(s) (1 - m)), what torch's
// int64 * python-float -> float32 -> copy_ into int64 computes, utils/model_utils.py:33-38)
__global__ void __launch_bounds__(256) ema_multi_kernel(const rcf_ema_chunk *__restrict__ tab, float m, float om, float om_counter) {
    const rcf_ema_chunk c = tab[blockIdx.x];
    if (c.kind == 0) {
        float *d = static_cast<float *>(c.dst);
        const float *s = static_cast<const float *>(c.src);
        for (long i = threadIdx.x; i < c.n; i += blockDim.x) d[i] = d[i] * m + s[i] * om;
    } else {
        long long *d = static_cast<long long *>(c.dst);
        const long long *s = static_cast<const long long *>(c.src);
        for (long i = threadIdx.x; i < c.n; i += blockDim.x) {
            const float a = (float)d[i] * m, b = (float)s[i] * om_counter;
            d[i] = (long long)(a + b);
        }
    }
}

__global__ void __launch_bounds__(256) fill_kernel(float *__restrict__ p, long n, float v) {
    const long step = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step) p[i] = v;
}
}  // namespace

extern "C" int rcf_adam_step_f32(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, long n, float lr,
                                 float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale,
                                 void *stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || n <= 0 || step < 1) return RCF_EINVAL;
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    hipLaunchKernelGGL(adam_kernel, dim3(ew_blocks(n)), dim3(256), 0, rcf_stream(stream), param, grad, exp_avg,
                       exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, (float)bc1, (float)sqrt(bc2), grad_scale);
    RCF_LAUNCH_CHECK();
    return 0;
}

extern "C" int rcf_ema_update_f32(float *dest, const float *src, long n, float m, void *stream) {
    if (!dest || !src || n <= 0) return RCF_EINVAL;
    hipLaunchKernelGGL(ema_kernel, dim3(ew_blocks(n)), dim3(256), 0, rcf_stream(stream), dest, src, n, m);
    RCF_LAUNCH_CHECK();
    return 0;
}

extern "C" int rcf_ema_update_multi(const rcf_ema_chunk *chunks_dev, int count, float m, float one_minus_m_counter, void *stream) {
    if (!chunks_dev || count <= 0) return RCF_EINVAL;
    // fp32 entries: 1.0f - m like rcf_ema_update_f32 (same bits as the per-tensor launches); counters: the caller's float(1.0 - m)
    hipLaunchKernelGGL(ema_multi_kernel, dim3((unsigned)count), dim3(256), 0, rcf_stream(stream), chunks_dev, m, 1.0f - m, one_minus_m_counter);
    RCF_LAUNCH_CHECK();
    return 0;
}

extern "C" int rcf_fill_f32(float *p, long n, float v, void *stream) {
    if (!p || n <= 0) return RCF_EINVAL;
    hipLaunchKernelGGL(fill_kernel, dim3(ew_blocks(n)), dim3(256), 0, rcf_stream(stream), p, n, v);
    RCF_LAUNCH_CHECK();
    return 0;
}

namespace {
// Philox4x32-10 (Salmon et al., SC'11: the counter-based generator torch's own CUDA dropout uses): counter = (element index,
// 0, 0, 0), key = the 64-bit seed.  One 32-bit word per element suffices here.
__device__ __forceinline__ unsigned philox_word(unsigned long long idx, unsigned long long seed) {
    unsigned c0 = (unsigned)idx, c1 = (unsigned)(idx >> 32), c2 = 0u, c3 = 0u;
    unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return c0;
}
__global__ void dropout2d_scale_kernel(float *__restrict__ out, long n, float p, unsigned long long seed) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float u = (float)(philox_word((unsigned long long)i, seed) >> 8) * (1.0f / 16777216.0f);      // [0, 1), 24 bits
    out[i] = u >= p ? 1.0f / (1.0f - p) : 0.0f;
}
}  // namespace

/* nn.Dropout2d's draw (models/decode_head.py:84-87): out[n][c] = 0 with probability p, 1 / (1 - p) otherwise -- the per-plane
 * scale the batch-norm apply pass multiplies in (rcf_bn_apply_mp chan_scale).  Counter-based: the same (seed, n * C) always gives
 * the same draw, whatever the launch geometry. */
extern "C" int rcf_dropout2d_scale_f32(float *out, long n, float p, unsigned long long seed, void *stream) {
    if (!out || n <= 0 || !(p >= 0.f && p < 1.f)) return RCF_EINVAL;
    hipLaunchKernelGGL(dropout2d_scale_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, rcf_stream(stream), out, n, p, seed);
    RCF_LAUNCH_CHECK();
    return 0;
}

extern "C" const char *rcf_version(void) { return "rcf_hip 0.1.0 gfx950"; }
