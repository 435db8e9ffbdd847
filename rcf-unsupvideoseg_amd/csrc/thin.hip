// Thin 1x1 convs: at most 16 output channels on a wide input -- the decode heads' classifiers (models/fcn_head.py conv_seg:
// 256 -> num_classes at 120 x 214, and the flow head's 256 -> 16).  As GEMMs they are a [pixels x Cin] x [Cin x <=16] product:
// 2 - 8 FLOP per byte, nothing for a matrix core to do; the implicit-GEMM kernels run them at 0.005 - 0.16 of their rooflines
// (a 64-wide tile with 4 live columns).  Here they are what they are -- streaming passes over the wide tensor with fp32 FMAs:
//   forward        y[p][n]  = sum_c x[p][c] w[n][c] (+ bias[n]) (+ y)      reads  x once, W from LDS
//   data gradient  dx[p][c] = sum_n dy[p][n] w[n][c] (+ dx)                 writes dx once
//   weight grad.   dw[n][c] (+)= sum_p dy[p][n] x[p][c]                    reads  x once; fixed-order two-level sum
// fp32 entry points only (rcf_conv2d_{fwd,dgrad,wgrad}_region_f32 hand over when rcf_thin_ok says so); exact products instead of
// three fp16 partial products: closer to float64 than the kernels they replace.  Deterministic.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rcf_common.h"

namespace {
__device__ __forceinline__ void block_amax_u(unsigned mx, unsigned *__restrict__ amax) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = max(mx, (unsigned)__shfl_xor((int)mx, o, 64));
    if ((threadIdx.x & 63) == 0 && mx > __hip_atomic_load(amax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(amax, mx);
}

// 16 lanes per pixel (lane l: channels 4 (l + 16 j) .. + 3, a coalesced 256-byte segment per j), 16 pixels per block and pass;
// a lane's J <= 8 quads of a pixel are all asked for before the first multiply
template <int NOUT>
__global__ void __launch_bounds__(256) thin_fwd_kernel(const float *__restrict__ x, int x_pitch, const float *__restrict__ w,
                                                       const float *__restrict__ bias, float *__restrict__ y, int y_pitch,
                                                       long rows, int Cin, int beta, unsigned *__restrict__ amax_y) {
    extern __shared__ float wl[];                      // [NOUT][Cin]
    for (int i = threadIdx.x; i < NOUT * Cin; i += 256) wl[i] = w[i];
    __syncthreads();
    const int l = threadIdx.x & 15, pp = threadIdx.x >> 4;
    const int J = Cin >> 6;
    unsigned mx = 0u;
    for (long p0 = (long)blockIdx.x * 16; p0 < rows; p0 += (long)gridDim.x * 16) {
        const long p = p0 + pp;
        float acc[NOUT];
#pragma unroll
        for (int n = 0; n < NOUT; ++n) acc[n] = 0.f;
        if (p < rows) {
            const float *xr = x + p * x_pitch + 4 * l;
            f32x4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (j < J) v[j] = *reinterpret_cast<const f32x4 *>(xr + 64 * j);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (j < J) {
                    const float *wj = wl + 4 * l + 64 * j;
#pragma unroll
                    for (int n = 0; n < NOUT; ++n) {
                        const f32x4 ww = *reinterpret_cast<const f32x4 *>(wj + n * Cin);
                        acc[n] = fmaf(v[j][0], ww[0], acc[n]);
                        acc[n] = fmaf(v[j][1], ww[1], acc[n]);
                        acc[n] = fmaf(v[j][2], ww[2], acc[n]);
                        acc[n] = fmaf(v[j][3], ww[3], acc[n]);
                    }
                }
            }
        }
#pragma unroll
        for (int n = 0; n < NOUT; ++n) {
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) acc[n] += __shfl_xor(acc[n], o, 16);
        }
        if (p < rows && l < NOUT / 4) {
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float t = 0.f;
#pragma unroll
                for (int n = 0; n < NOUT; ++n) t = (n == 4 * l + e) ? acc[n] : t;      // (static indexing: the accumulators stay in registers)
                o[e] = t + (bias ? bias[4 * l + e] : 0.f);
            }
            float *yr = y + p * y_pitch + 4 * l;
            if (beta) o += *reinterpret_cast<const f32x4 *>(yr);
            *reinterpret_cast<f32x4 *>(yr) = o;
#pragma unroll
            for (int e = 0; e < 4; ++e) mx = max(mx, __float_as_uint(fabsf(o[e])));
        }
    }
    if (amax_y) block_amax_u(mx, amax_y);
}

// a thread owns one channel quad of a pixel: Cin / 4 threads per pixel, 1024 / Cin pixels per block and pass
template <int NOUT>
__global__ void __launch_bounds__(256) thin_dgrad_kernel(const float *__restrict__ dy, int dy_pitch, const float *__restrict__ w,
                                                         float *__restrict__ dx, int dx_pitch, long rows, int Cin, int beta,
                                                         unsigned *__restrict__ amax_y) {
    extern __shared__ float wl[];                      // [NOUT][Cin]
    for (int i = threadIdx.x; i < NOUT * Cin; i += 256) wl[i] = w[i];
    __syncthreads();
    const int Q = Cin >> 2, ppb = 256 / Q;
    const int cq = threadIdx.x % Q, pp = threadIdx.x / Q;
    f32x4 wr[NOUT];
#pragma unroll
    for (int n = 0; n < NOUT; ++n) wr[n] = *reinterpret_cast<const f32x4 *>(wl + n * Cin + 4 * cq);
    unsigned mx = 0u;
    for (long p = (long)blockIdx.x * ppb + pp; p < rows; p += (long)gridDim.x * ppb) {
        const float *g = dy + p * dy_pitch;
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int n4 = 0; n4 < NOUT; n4 += 4) {
            const f32x4 gv = *reinterpret_cast<const f32x4 *>(g + n4);      // the same 16 bytes for every thread of the pixel
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                o[0] = fmaf(gv[e], wr[n4 + e][0], o[0]);
                o[1] = fmaf(gv[e], wr[n4 + e][1], o[1]);
                o[2] = fmaf(gv[e], wr[n4 + e][2], o[2]);
                o[3] = fmaf(gv[e], wr[n4 + e][3], o[3]);
            }
        }
        float *d = dx + p * dx_pitch + 4 * cq;
        if (beta) o += *reinterpret_cast<const f32x4 *>(d);
        *reinterpret_cast<f32x4 *>(d) = o;
#pragma unroll
        for (int e = 0; e < 4; ++e) mx = max(mx, __float_as_uint(fabsf(o[e])));
    }
    if (amax_y) block_amax_u(mx, amax_y);
}

// the same ownership, four pixels in flight per thread; the block's pixel lanes are added in lane order through LDS and every
// block leaves ONE partial [NOUT][Cin], summed in a fixed order by thin_wsum_kernel
template <int NOUT>
__global__ void __launch_bounds__(256) thin_wgrad_kernel(const float *__restrict__ x, int x_pitch, const float *__restrict__ dy,
                                                         int dy_pitch, float *__restrict__ partial, long rows, int Cin) {
    extern __shared__ float red[];                     // [NOUT][Cin]
    const int Q = Cin >> 2, ppb = 256 / Q;
    const int cq = threadIdx.x % Q, pp = threadIdx.x / Q;
    f32x4 acc[NOUT];
#pragma unroll
    for (int n = 0; n < NOUT; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    const long step = (long)gridDim.x * ppb;
    auto take = [&](const f32x4 xv, const float *g) {
#pragma unroll
        for (int n4 = 0; n4 < NOUT; n4 += 4) {
            const f32x4 gv = *reinterpret_cast<const f32x4 *>(g + n4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[n4 + e][0] = fmaf(gv[e], xv[0], acc[n4 + e][0]);
                acc[n4 + e][1] = fmaf(gv[e], xv[1], acc[n4 + e][1]);
                acc[n4 + e][2] = fmaf(gv[e], xv[2], acc[n4 + e][2]);
                acc[n4 + e][3] = fmaf(gv[e], xv[3], acc[n4 + e][3]);
            }
        }
    };
    long p = (long)blockIdx.x * ppb + pp;
    for (; p + 3 * step < rows; p += 4 * step) {
        f32x4 xv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) xv[u] = *reinterpret_cast<const f32x4 *>(x + (p + u * step) * x_pitch + 4 * cq);
#pragma unroll
        for (int u = 0; u < 4; ++u) take(xv[u], dy + (p + u * step) * dy_pitch);
    }
    for (; p < rows; p += step) take(*reinterpret_cast<const f32x4 *>(x + p * x_pitch + 4 * cq), dy + p * dy_pitch);
    for (int r = 0; r < ppb; ++r) {                     // lane 0 stores, lanes 1 .. ppb - 1 add in order
        if (pp == r) {
#pragma unroll
            for (int n = 0; n < NOUT; ++n) {
                f32x4 *d = reinterpret_cast<f32x4 *>(red + n * Cin + 4 * cq);
                *d = r == 0 ? acc[n] : *d + acc[n];
            }
        }
        __syncthreads();
    }
    float *dst = partial + (long)blockIdx.x * NOUT * Cin;
    for (int i = threadIdx.x; i < NOUT * Cin; i += 256) dst[i] = red[i];
}

// dw[i] (+)= sum_k partial[k][i], four interleaved chains in fp64 combined in a fixed order
__global__ void __launch_bounds__(256) thin_wsum_kernel(const float *__restrict__ partial, int K, long n, float *__restrict__ dw, int beta) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    int k = 0;
    for (; k + 3 < K; k += 4) {
        a0 += (double)partial[(long)k * n + i];
        a1 += (double)partial[(long)(k + 1) * n + i];
        a2 += (double)partial[(long)(k + 2) * n + i];
        a3 += (double)partial[(long)(k + 3) * n + i];
    }
    for (; k < K; ++k) a0 += (double)partial[(long)k * n + i];
    const double t = (a0 + a1) + (a2 + a3);
    dw[i] = (float)(beta ? (double)dw[i] + t : t);
}

inline int thin_blocks(long rows, int ppb) {
    long b = (rows + ppb - 1) / ppb;
    b = (b + 7) / 8;                                   // >= 8 passes per block
    if (b > 1024) b = 1024;
    return (int)(b < 1 ? 1 : b);
}
}  // namespace

// 1x1, stride 1, no padding, 4 / 8 / 16 output channels, 64 .. 512 input channels (a power of two times 64), fp32 tensors
bool rcf_thin_ok(const rcf_conv_shape *s) {
    if (s->R != 1 || s->S != 1 || s->stride != 1 || s->pad != 0 || s->H != s->Ho || s->W != s->Wo) return false;
    if (s->Cout != 4 && s->Cout != 8 && s->Cout != 16) return false;
    if (s->Cin != 64 && s->Cin != 128 && s->Cin != 256 && s->Cin != 512) return false;
    if (s->flags & (RCF_CONV_X_PLANES | RCF_CONV_DY_PLANES)) return false;
    if (s->x_pitch % 4 || s->y_pitch % 4 || s->x_pitch < s->Cin || s->y_pitch < s->Cout) return false;
    return true;
}

inline int thin_wgrad_blocks(long rows, int ppb) {
    long b = (rows + ppb - 1) / ppb / 32;              // >= 32 pixels per thread; two workgroups per CU at most
    if (b > 512) b = 512;
    return (int)(b < 1 ? 1 : b);
}

size_t rcf_thin_wgrad_workspace_bytes(const rcf_conv_shape *s) {
    const long rows = (long)s->N * s->H * s->W;
    return (size_t)thin_wgrad_blocks(rows, 1024 / s->Cin) * s->Cout * s->Cin * sizeof(float);
}

#define RCF_THIN_DISPATCH(NOUTv, CALL) \
    switch (NOUTv) {                   \
        case 4: { constexpr int NO = 4; CALL; } break;   \
        case 8: { constexpr int NO = 8; CALL; } break;   \
        default: { constexpr int NO = 16; CALL; } break; \
    }

int rcf_thin_fwd(const float *x, const float *w, const float *bias, float *y, const rcf_conv_shape *s, int beta, hipStream_t st) {
    const long rows = (long)s->N * s->H * s->W;
    const size_t lds = (size_t)s->Cout * s->Cin * sizeof(float);
    RCF_THIN_DISPATCH(s->Cout, hipLaunchKernelGGL(thin_fwd_kernel<NO>, dim3(thin_blocks(rows, 16)), dim3(256), lds, st, x, s->x_pitch, w,
                                                  bias, y, s->y_pitch, rows, s->Cin, beta, s->amax_y));
    RCF_LAUNCH_CHECK();
    return 0;
}

int rcf_thin_dgrad(const float *dy, const float *w, float *dx, const rcf_conv_shape *s, int beta, hipStream_t st) {
    const long rows = (long)s->N * s->H * s->W;
    const size_t lds = (size_t)s->Cout * s->Cin * sizeof(float);
    const int ppb = 1024 / s->Cin;
    RCF_THIN_DISPATCH(s->Cout, hipLaunchKernelGGL(thin_dgrad_kernel<NO>, dim3(thin_blocks(rows, ppb)), dim3(256), lds, st, dy, s->y_pitch,
                                                  w, dx, s->x_pitch, rows, s->Cin, beta, s->amax_y));
    RCF_LAUNCH_CHECK();
    return 0;
}

int rcf_thin_wgrad(const float *x, const float *dy, float *dw, const rcf_conv_shape *s, int beta, void *workspace, size_t workspace_bytes,
                   hipStream_t st) {
    if (!workspace || workspace_bytes < rcf_thin_wgrad_workspace_bytes(s)) return RCF_EWORKSPACE;
    const long rows = (long)s->N * s->H * s->W;
    const int ppb = 1024 / s->Cin, blocks = thin_wgrad_blocks(rows, ppb);
    const size_t lds = (size_t)s->Cout * s->Cin * sizeof(float);
    RCF_THIN_DISPATCH(s->Cout, hipLaunchKernelGGL(thin_wgrad_kernel<NO>, dim3(blocks), dim3(256), lds, st, x, s->x_pitch, dy, s->y_pitch,
                                                  (float *)workspace, rows, s->Cin));
    RCF_LAUNCH_CHECK();
    const long n = (long)s->Cout * s->Cin;
    hipLaunchKernelGGL(thin_wsum_kernel, dim3(rcf_cdiv(n, 256)), dim3(256), 0, st, (const float *)workspace, blocks, n, dw, beta);
    RCF_LAUNCH_CHECK();
    return 0;
}
