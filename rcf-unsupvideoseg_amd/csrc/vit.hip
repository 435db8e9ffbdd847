// Small kernels of the DINO ViT forward and the soft-NCut refinement:
// models/dino_vit.py:110-167 (LayerNorm eps 1e-6, softmax(q k^T * scale), GELU lives in the GEMM epilogue),
// tools/SemanticConstraintsAndMAA/semantic_constraints.py:21-75 (soft NCut value / Adam refinement on a mask).
// The matrix products run on the split-bf16 conv kernel (rcf_gemm_nt_f32).  HBM-bound helpers, wave64.
#include "rcf_common.h"

namespace {

__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max_f(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// y[r][c] = (x[r][c] - mean_r) * rstd_r * gamma[c] + beta[c]; one wavefront per row, two-pass in registers (C <= 2048)
__global__ void __launch_bounds__(256) layernorm_kernel(const float *__restrict__ x, int x_pitch, float *__restrict__ y,
                                                        int y_pitch, long rows, int C, const float *__restrict__ gamma,
                                                        const float *__restrict__ beta, float eps,
                                                        unsigned *__restrict__ amax_out) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    unsigned omax = 0u;
    const float *xr = x + r * x_pitch;
    const int nq = C / 4;                         // float4 per row; lane handles quads lane, lane+64, ... (<= 8)
    f32x4 q[8];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int i = lane + 64 * k;
        q[k] = i < nq ? *reinterpret_cast<const f32x4 *>(xr + i * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        s += (q[k][0] + q[k][1]) + (q[k][2] + q[k][3]);
    }
    const float mean = wave_sum_f(s) / (float)C;
    float s2 = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k)
        if (lane + 64 * k < nq)
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = q[k][e] - mean; s2 += d * d; }
    const float rstd = 1.0f / sqrtf(wave_sum_f(s2) / (float)C + eps);      // biased variance, like nn.LayerNorm
    float *yr = y + r * y_pitch;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int i = lane + 64 * k;
        if (i >= nq) continue;
        const f32x4 g = *reinterpret_cast<const f32x4 *>(gamma + i * 4), b = *reinterpret_cast<const f32x4 *>(beta + i * 4);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            o[e] = (q[k][e] - mean) * rstd * g[e] + b[e];
            omax = max(omax, __float_as_uint(fabsf(o[e])));
        }
        *reinterpret_cast<f32x4 *>(yr + i * 4) = o;
    }
    if (amax_out) {            // the range of the GEMM that reads y (rcf_gemm_nt_f32's amax_a)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) omax = max(omax, (unsigned)__shfl_xor((int)omax, o));
        if (lane == 0 && omax > __hip_atomic_load(amax_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(amax_out, omax);
    }
}

// in place: row r <- softmax(scale * row r) over the first n columns; columns [n, pitch) are zeroed (K padding of the
// following P.V product).  One workgroup per row.  Rows of up to 256*4*QMAX floats with a 16-byte aligned pitch are
// read ONCE into registers (float4 per lane) and written once; longer / unaligned rows take the three-pass path.
constexpr int SM_QMAX = 8;                         // 8192 columns
__global__ void __launch_bounds__(256) softmax_rows_kernel(float *__restrict__ s, long pitch, int n, float scale) {
    __shared__ float red[4];
    float *row = s + (long)blockIdx.x * pitch;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const bool fast = (pitch % 4 == 0) && pitch <= 256L * 4 * SM_QMAX;
    if (fast) {
        const int nq = (int)(pitch / 4);
        f32x4 q[SM_QMAX];
        float mx = -INFINITY;
#pragma unroll
        for (int k = 0; k < SM_QMAX; ++k) {
            const int i = threadIdx.x + 256 * k;
            if (i < nq) {
                q[k] = *reinterpret_cast<const f32x4 *>(row + i * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    q[k][e] = (i * 4 + e < n) ? q[k][e] * scale : -INFINITY;
                    mx = fmaxf(mx, q[k][e]);
                }
            }
        }
        mx = wave_max_f(mx);
        if (lane == 0) red[wv] = mx;
        __syncthreads();
        mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        __syncthreads();
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < SM_QMAX; ++k)
            if (threadIdx.x + 256 * k < nq)
#pragma unroll
                for (int e = 0; e < 4; ++e) { q[k][e] = expf(q[k][e] - mx); sum += q[k][e]; }   // exp(-inf) = 0 on the padding
        sum = wave_sum_f(sum);
        if (lane == 0) red[wv] = sum;
        __syncthreads();
        const float inv = 1.0f / ((red[0] + red[1]) + (red[2] + red[3]));
#pragma unroll
        for (int k = 0; k < SM_QMAX; ++k) {
            const int i = threadIdx.x + 256 * k;
            if (i < nq) *reinterpret_cast<f32x4 *>(row + i * 4) = q[k] * inv;
        }
        return;
    }
    float mx = -INFINITY;
    for (int i = threadIdx.x; i < n; i += 256) mx = fmaxf(mx, row[i] * scale);
    mx = wave_max_f(mx);
    if (lane == 0) red[wv] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float sum = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float e = expf(row[i] * scale - mx);
        row[i] = e;
        sum += e;
    }
    sum = wave_sum_f(sum);
    if (lane == 0) red[wv] = sum;
    __syncthreads();
    const float inv = 1.0f / ((red[0] + red[1]) + (red[2] + red[3]));
    for (int i = threadIdx.x; i < n; i += 256) row[i] *= inv;
    for (int i = n + threadIdx.x; i < pitch; i += 256) row[i] = 0.f;
}

// dst[c][r] = src[r][c] for r < rows, c < cols; dst columns [rows, dpitch) zero-filled
__global__ void __launch_bounds__(256) transpose2d_kernel(const float *__restrict__ src, long spitch,
                                                          float *__restrict__ dst, long dpitch, int rows, int cols) {
    __shared__ float tile[32][33];
    const int r0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8) {
        const int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < rows && c < cols) ? src[(long)r * spitch + c] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, r = r0 + tx;
        if (c < cols && r < dpitch) dst[(long)c * dpitch + r] = tile[tx][i];
    }
}

// rows of x scaled to unit L2 norm (F.normalize(p=2, dim=1), eps 1e-12); one wavefront per row
__global__ void __launch_bounds__(256) l2_normalize_rows_kernel(const float *__restrict__ x, long x_pitch,
                                                                float *__restrict__ y, long y_pitch, long rows, int C) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    float s = 0.f;
    for (int i = lane; i < C; i += 64) { const float v = x[r * x_pitch + i]; s += v * v; }
    const float nrm = fmaxf(sqrtf(wave_sum_f(s)), 1e-12f);
    for (int i = lane; i < C; i += 64) y[r * y_pitch + i] = x[r * x_pitch + i] / nrm;
}

// a[i][j] = g[i][j] > tau ? 1 : eps   (in place on the Gram matrix; columns >= n untouched)
__global__ void __launch_bounds__(256) affinity_threshold_kernel(float *__restrict__ g, long pitch, int n, float tau,
                                                                 float eps) {
    const long total = (long)n * n;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / n;
        const int c = (int)(i - r * n);
        float *p = g + r * pitch + c;
        *p = *p > tau ? 1.0f : eps;
    }
}

// u = A x (one workgroup per row, fp64 accumulation), s = A 1 optional
__global__ void __launch_bounds__(256) matvec_kernel(const float *__restrict__ a, long pitch, int n,
                                                     const float *__restrict__ x, double *__restrict__ u,
                                                     double *__restrict__ rowsum) {
    __shared__ double red[2][4];
    const float *row = a + (long)blockIdx.x * pitch;
    double su = 0, ss = 0;
    for (int i = threadIdx.x; i < n; i += 256) {
        const double v = row[i];
        su += v * (double)x[i];
        ss += v;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { su += __shfl_xor(su, o, 64); ss += __shfl_xor(ss, o, 64); }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) { red[0][wv] = su; red[1][wv] = ss; }
    __syncthreads();
    if (threadIdx.x == 0) {
        u[blockIdx.x] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        if (rowsum) rowsum[blockIdx.x] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
}

// NCut = cut/a + cut/(S-a) with cut = s.x - x.u, a = s.x, S = sum s;  grad_i = (s_i - 2 u_i)(1/a + 1/(S-a))
// - cut s_i / a^2 + cut s_i / (S-a)^2.  One workgroup; out[0] = NCut.
__global__ void __launch_bounds__(256) ncut_value_grad_kernel(const float *__restrict__ x, const double *__restrict__ u,
                                                              const double *__restrict__ s, int n,
                                                              float *__restrict__ grad, float *__restrict__ out) {
    __shared__ double red[3][4];
    __shared__ double tot[3];
    double sx = 0, xu = 0, S = 0;
    for (int i = threadIdx.x; i < n; i += 256) {
        sx += s[i] * (double)x[i];
        xu += (double)x[i] * u[i];
        S += s[i];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { sx += __shfl_xor(sx, o, 64); xu += __shfl_xor(xu, o, 64); S += __shfl_xor(S, o, 64); }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) { red[0][wv] = sx; red[1][wv] = xu; red[2][wv] = S; }
    __syncthreads();
    if (threadIdx.x < 3) tot[threadIdx.x] = (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
    __syncthreads();
    const double a = tot[0], cut = tot[0] - tot[1], b = tot[2] - tot[0];
    if (threadIdx.x == 0 && out) out[0] = (float)(cut / a + cut / b);
    if (grad) {
        const double c1 = 1.0 / a + 1.0 / b, c2 = cut / (a * a), c3 = cut / (b * b);
        for (int i = threadIdx.x; i < n; i += 256) grad[i] = (float)((s[i] - 2.0 * u[i]) * c1 - c2 * s[i] + c3 * s[i]);
    }
}

__global__ void __launch_bounds__(256) clamp01_kernel(float *__restrict__ x, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] = fminf(fmaxf(x[i], 0.f), 1.f);
}

}  // namespace

extern "C" int rcf_layernorm_f32(const float *x, int x_pitch, float *y, int y_pitch, long rows, int C,
                                 const float *gamma, const float *beta, float eps, unsigned *amax_out, void *stream) {
    if (!x || !y || !gamma || !beta || rows <= 0 || C <= 0 || C % 4 || C > 2048 || x_pitch % 4 || y_pitch % 4) return RCF_EINVAL;
    hipLaunchKernelGGL(layernorm_kernel, dim3((unsigned)rcf_cdiv(rows, 4)), dim3(256), 0, rcf_stream(stream), x, x_pitch, y,
                       y_pitch, rows, C, gamma, beta, eps, amax_out);
    RCF_LAUNCH_CHECK();
    return 0;
}

extern "C" int rcf_softmax_rows_f32(float *s, long pitch, long rows, int n, float scale, void *stream) {
    if (!s || rows <= 0 || n <= 0 || pitch < n) return RCF_EINVAL;
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)rows), dim3(256), 0, rcf_stream(stream), s, pitch, n, scale);
    RCF_LAUNCH_CHECK();
    return 0;
}

extern "C" int rcf_transpose2d_f32(const float *src, long spitch, float *dst, long dpitch, int rows, int cols,
                                   void *stream) {
    if (!src || !dst || rows <= 0 || cols <= 0 || spitch < cols || dpitch < rows) return RCF_EINVAL;
    hipLaunchKernelGGL(transpose2d_kernel, dim3(rcf_cdiv(dpitch, 32), rcf_cdiv(cols, 32)), dim3(256), 0, rcf_stream(stream),
                       src, spitch, dst, dpitch, rows, cols);
    RCF_LAUNCH_CHECK();
    return 0;
}

extern "C" int rcf_l2_normalize_rows_f32(const float *x, long x_pitch, float *y, long y_pitch, long rows, int C,
                                         void *stream) {
    if (!x || !y || rows <= 0 || C <= 0) return RCF_EINVAL;
    hipLaunchKernelGGL(l2_normalize_rows_kernel, dim3((unsigned)rcf_cdiv(rows, 4)), dim3(256), 0, rcf_stream(stream), x,
                       x_pitch, y, y_pitch, rows, C);
    RCF_LAUNCH_CHECK();
    return 0;
}

extern "C" int rcf_affinity_threshold_f32(float *gram, long pitch, int n, float tau, float eps, void *stream) {
    if (!gram || n <= 0 || pitch < n) return RCF_EINVAL;
    hipLaunchKernelGGL(affinity_threshold_kernel, dim3(4096), dim3(256), 0, rcf_stream(stream), gram, pitch, n, tau, eps);
    RCF_LAUNCH_CHECK();
    return 0;
}

/* one soft-NCut evaluation on a mask x[n] against the affinity a[n][n]: u / rowsum are fp64 scratch [n];
 * value_out[0] = NCut (may be NULL), grad[n] = d NCut / d x (may be NULL).  rowsum is (re)computed when
 * `compute_rowsum` is set (it only depends on the affinity). */
extern "C" int rcf_ncut_value_grad_f32(const float *affinity, long pitch, int n, const float *x, double *u,
                                       double *rowsum, int compute_rowsum, float *grad, float *value_out,
                                       void *stream) {
    if (!affinity || !x || !u || !rowsum || n <= 0 || pitch < n) return RCF_EINVAL;
    hipStream_t st = rcf_stream(stream);
    hipLaunchKernelGGL(matvec_kernel, dim3(n), dim3(256), 0, st, affinity, pitch, n, x, u, compute_rowsum ? rowsum : nullptr);
    hipLaunchKernelGGL(ncut_value_grad_kernel, dim3(1), dim3(256), 0, st, x, (const double *)u, (const double *)rowsum, n,
                       grad, value_out);
    RCF_LAUNCH_CHECK();
    return 0;
}

extern "C" int rcf_clamp01_f32(float *x, int n, void *stream) {
    if (!x || n <= 0) return RCF_EINVAL;
    hipLaunchKernelGGL(clamp01_kernel, dim3(rcf_cdiv(n, 256)), dim3(256), 0, rcf_stream(stream), x, n);
    RCF_LAUNCH_CHECK();
    return 0;
}
