// 1x1 conv -> training-mode batch norm (-> residual add -> ReLU) without the conv's output in memory -- the bf16 step's
// bottleneck tail conv3 -> bn3 -> (+ identity) -> ReLU (models/resnet.py:281-296) and the 1x1 downsample conv -> norm
// (models/res_layer.py:53-63).  A 1x1 conv is LINEAR in its input, z[p][c] = sum_k x[p][k] w[c][k], so everything the batch
// norm needs from z is a function of two small moments of x:
//     A1[k]   = sum_p x[p][k]                       (column sums)
//     S[j][k] = sum_p x[p][j] x[p][k]               (Gram matrix, K x K; the weight-gradient kernel on (x, x))
//   forward:   sum_p z[p][c]   = A1 . w[c]          sum_p z[p][c]^2 = w[c]^T S w[c] = w[c] . P[c],   P = W S
//              -> mean, invstd as always; the conv kernel then applies y = relu(z * scale + shift + residual) in its epilogue
//                 (rcf_conv2d_fwd_affine_bf16) and z never exists.
//   backward (g = dy under the ReLU mask, G[c][k] = sum_p g[p][c] x[p][k] -- the weight-gradient kernel on (x, g)):
//              sum g zhat = invstd (w[c] . G[c] - mean sum g)           (the norm's second backward sum)
//              dW[c][k]   = a_c (G - m_c A1[k] - q_c invstd_c (P[c][k] - mean_c A1[k]))        a = gamma invstd, m = sum g / n,
//              dx         = g Wg^T - x T + c0,   Wg = a W,  T = W^T diag(a invstd q) W,         q = sum g zhat / n
//                           c0[k] = sum_c (a mean invstd q - a m)_c w[c][k]
//              -> two conv launches on tensors that exist anyway (g, x); dz never exists.
// Same mathematics as BatchNorm's autograd (checked against it in float64: tests/test_fold_cpu.py restates this file in torch);
// what changes is the rounding: the norm sees the fp32 accumulators instead of their bf16 roundings.
// W everywhere below is the weight AS THE bf16 KERNELS SEE IT: the fp32 master copy rounded to bf16.
#include "rcf_common.h"

namespace {

__device__ __forceinline__ float wr(float w) { return (float)(bf16_t)w; }

// C_z[M][N] = sum over the inner range of split z of A(m, i) B(i, n): the small fp32 products of the fold (P = W S; T = W^T Wd).
// A(m, i) = A[m sam + i sai] (one of the strides is 1), rounded to bf16 when ROUND_A (A is the master weight); B(i, n) = B[i sbi + n]
// 64 x 64 tile per workgroup, 4 x 4 outputs per thread, inner steps of 16 through LDS.  M, N % 64 == 0, inner per split % 16 == 0.
// CENTER_B: B(i, n) = B[i sbi + n] - cvec[i] cvec[n] cinv, formed in fp64 as the tile is loaded -- P = W (S - A1 A1^T / n): the
// Gram matrix is centred BEFORE the fp32 contraction (the variance as w^T S w / n - mean^2 loses
// mean^2 / var of its digits in the K-long fp32 chain; after centring the chain's rounding is relative to the variance itself)
template <bool ROUND_A, bool CENTER_B = false>
__global__ void __launch_bounds__(256) small_gemm_kernel(const float *__restrict__ A, long sam, long sai, const float *__restrict__ B,
                                                         long sbi, float *__restrict__ C, int M, int N, int inner_per_split,
                                                         const double *__restrict__ cvec = nullptr, double cinv = 0.0) {
    __shared__ __attribute__((aligned(16))) float As[16][68], Bs[16][68];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const int i0 = blockIdx.z * inner_per_split, i1 = i0 + inner_per_split;
    float acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = 0.f;
    for (int ib = i0; ib < i1; ib += 16) {
        if (sai == 1) {                                   // rows of A contiguous along the inner index: thread = (m, 4 inner)
            const int m = tid >> 2, iq = (tid & 3) * 4;
            const f32x4 v = *reinterpret_cast<const f32x4 *>(A + (long)(m0 + m) * sam + ib + iq);
#pragma unroll
            for (int e = 0; e < 4; ++e) As[iq + e][m] = ROUND_A ? (float)(bf16_t)v[e] : v[e];
        } else {                                          // contiguous along m: thread = (inner, 4 m)
            const int i = tid >> 4, mq = (tid & 15) * 4;
            const f32x4 v = *reinterpret_cast<const f32x4 *>(A + (long)(ib + i) * sai + m0 + mq);
#pragma unroll
            for (int e = 0; e < 4; ++e) As[i][mq + e] = ROUND_A ? (float)(bf16_t)v[e] : v[e];
        }
        {
            const int i = tid >> 4, nq = (tid & 15) * 4;
            f32x4 bv = *reinterpret_cast<const f32x4 *>(B + (long)(ib + i) * sbi + n0 + nq);
            if constexpr (CENTER_B) {
                const double ci = cvec[ib + i] * cinv;
#pragma unroll
                for (int e = 0; e < 4; ++e) bv[e] = (float)((double)bv[e] - ci * cvec[n0 + nq + e]);
            }
            *reinterpret_cast<f32x4 *>(&Bs[i][nq]) = bv;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const f32x4 a = *reinterpret_cast<const f32x4 *>(&As[i][ty * 4]);
            const f32x4 b = *reinterpret_cast<const f32x4 *>(&Bs[i][tx * 4]);
#pragma unroll
            for (int x = 0; x < 4; ++x)
#pragma unroll
                for (int y = 0; y < 4; ++y) acc[x][y] = fmaf(a[x], b[y], acc[x][y]);
        }
        __syncthreads();
    }
    float *Cz = C + (long)blockIdx.z * M * N;
#pragma unroll
    for (int x = 0; x < 4; ++x)
        *reinterpret_cast<f32x4 *>(Cz + (long)(m0 + ty * 4 + x) * N + n0 + tx * 4) = f32x4{acc[x][0], acc[x][1], acc[x][2], acc[x][3]};
}

// splits of the inner dimension so that about 256 workgroups run (each at least 64 deep)
inline int gemm_splits(int M, int N, int inner) {
    const int tiles = (M / 64) * (N / 64);
    int z = 1;
    while (tiles * z < 256 && inner / (2 * z) >= 64 && (inner / (2 * z)) % 16 == 0) z *= 2;
    return z;
}

// P = sum of its Z partial products, the CENTRED P_c = W (S - A1 A1^T / n_local) (written back as the final P[N][K], followed by the
// N local means of z, mloc = A1 . W[c] / n_local: what the backward pass needs to re-centre P on the global mean);
// sums[c] = A1 . W[c], sums[N + c] = the UNcentred sum of z^2 = W[c] . P_c[c] + (A1 . W[c])^2 / n_local, put together in fp64 (what
// SyncBN all-reduces); with `fin` the channel is finalized right here (local statistics: var = W[c] . P_c[c] / n, no subtraction).
// One wavefront per channel.
struct FoldFin {
    double count;
    float eps, momentum;
    const float *gamma, *beta;
    float *mean, *invstd, *scale, *shift, *rmean, *rvar;
    long long *nbt;
    int on;
};
// centred: `szz` is sum (z - mean)^2 already (the local path); otherwise the plain sum of squares (after an all-reduce)
__device__ __forceinline__ void fold_finalize_channel(const FoldFin &f, int c, double sz, double szz, bool centred = false) {
    const double m = sz / f.count;
    double var = centred ? szz / f.count : szz / f.count - m * m;
    if (var < 0) var = 0;
    const float mf = (float)m, is = (float)(1.0 / sqrt(var + (double)f.eps));
    f.mean[c] = mf;
    f.invstd[c] = is;
    const float a = f.gamma[c] * is;
    f.scale[c] = a;
    f.shift[c] = f.beta[c] - mf * a;
    if (f.rmean) f.rmean[c] = (1.f - f.momentum) * f.rmean[c] + f.momentum * mf;
    if (f.rvar) {
        const double unbiased = f.count > 1 ? var * (f.count / (f.count - 1.0)) : var;
        f.rvar[c] = (1.f - f.momentum) * f.rvar[c] + f.momentum * (float)unbiased;
    }
}
__global__ void __launch_bounds__(256) fold_stats_kernel(const float *__restrict__ Ppart, int Z, const double *__restrict__ A1,
                                                         const float *__restrict__ W, float *__restrict__ P,
                                                         double *__restrict__ sums, int N, int K, FoldFin fin, double rows_local) {
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (c == 0 && lane == 0 && fin.on && fin.nbt) *fin.nbt += 1;
    if (c >= N) return;
    double sz = 0, szz = 0;
    for (int k = lane; k < K; k += 64) {
        float pv = 0.f;
        for (int z = 0; z < Z; ++z) pv += Ppart[((long)z * N + c) * K + k];
        P[(long)c * K + k] = pv;
        const double w = (double)wr(W[(long)c * K + k]);
        sz += A1[k] * w;
        szz += w * (double)pv;
    }
    sz = wave_sum_d(sz);
    szz = wave_sum_d(szz);
    if (lane == 0) {
        P[(long)N * K + c] = (float)(sz / rows_local);                 // mloc: with local statistics exactly the mean below
        if (sums) { sums[c] = sz; sums[N + c] = szz + sz * sz / rows_local; }
        if (fin.on) fold_finalize_channel(fin, c, sz, szz, true);
    }
}

// bn_finalize_kernel's arithmetic (csrc/bn.hip) plus the folded constants scale = gamma invstd, shift = beta - mean scale: the
// SyncBN form (the sums were all-reduced between fold_stats_kernel and this)
__global__ void fold_finalize_kernel(const double *__restrict__ sums, int C, FoldFin fin) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c == 0 && fin.nbt) *fin.nbt += 1;
    if (c >= C) return;
    fold_finalize_channel(fin, c, sums[c], sums[C + c]);
}

// g = y > 0 ? dy : 0 (bf16, 8 channels per thread) and the column sums of g: partial[chunk][2C] (second half zero: the layout
// rcf_sum_partials_bn adds up).  Thread = (row phase rp, channel group cg); a workgroup walks rows [chunk * per, + per).
__global__ void __launch_bounds__(256) relu_mask_colsum_kernel(const bf16_t *dy, int dy_pitch, const bf16_t *__restrict__ y,
                                                               int y_pitch, bf16_t *g, int g_pitch, long rows, int C, long per,
                                                               double *__restrict__ partial) {
    __shared__ float red[256][8];
    const int groups = C >> 3;                             // <= 256
    const int rpp = 256 / groups;                          // rows per pass
    const int tid = threadIdx.x, cg = tid % groups, rp = tid / groups;
    const long r0 = (long)blockIdx.x * per, r1 = min(rows, r0 + per);
    float s[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = 0.f;
    if (rp < rpp) {
        for (long r = r0 + rp; r < r1; r += rpp) {
            const fvec<8> d = ldv<bf16_t, 8>(dy + r * dy_pitch + 8 * cg), m = ldv<bf16_t, 8>(y + r * y_pitch + 8 * cg);
            fvec<8> o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float v = m.q[e >> 2][e & 3] > 0.f ? d.q[e >> 2][e & 3] : 0.f;
                o.q[e >> 2][e & 3] = v;
                s[e] += v;
            }
            stv<bf16_t, 8>(g + r * g_pitch + 8 * cg, o);
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[tid][e] = s[e];
    __syncthreads();
    for (int i = tid; i < C; i += 256) {
        const int gq = i >> 3, e = i & 7;
        double t = 0;
        for (int p = 0; p < rpp; ++p) t += (double)red[p * groups + gq][e];
        double *o = partial + (long)blockIdx.x * 2 * C;
        o[i] = t;
        o[C + i] = 0.0;
    }
}

// sums2[c] = sum g;  sums2[N + c] = invstd (W[c] . G[c] - mean sum g).  One wavefront per channel.
__global__ void __launch_bounds__(256) fold_bwd_sums_kernel(const float *__restrict__ G, const float *__restrict__ W,
                                                            const double *__restrict__ colsums, const float *__restrict__ mean,
                                                            const float *__restrict__ invstd, double *__restrict__ sums2, int N,
                                                            int K) {
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (c >= N) return;
    double u = 0;
    for (int k = lane; k < K; k += 64) u += (double)wr(W[(long)c * K + k]) * (double)G[(long)c * K + k];
    u = wave_sum_d(u);
    if (lane == 0) {
        const double sg = colsums[c];
        sums2[c] = sg;
        sums2[N + c] = (double)invstd[c] * (u - (double)mean[c] * sg);
    }
}

// Wg^T = (a W)^T as the data gradient's bf16 weight operand (rows = k, K-step major over c): needs only the forward constants, so
// the data gradient g Wg^T can start before G exists
__global__ void __launch_bounds__(256) fold_wg_kernel(const float *__restrict__ W, const float *__restrict__ scale,
                                                      bf16_t *__restrict__ wg_t, int N, int K) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)N * K) return;
    const int c = (int)(i / K), k = (int)(i - (long)c * K);
    wg_t[((long)(c >> 5) * K + k) * 32 + (c & 31)] = (bf16_t)(scale[c] * wr(W[i]));
}

// per (c, k): dW += a (G - m A1 - q invstd (P - mean A1)), with P - mean A1 = P_c - (mean - mloc) A1 from the centred P_c and the local
// means rcf_fold_fwd_f32 left behind it (mean == mloc bit for bit with local statistics);  Wd = a invstd q W (fp32, for T);  c0 partial sums over this
// workgroup's PREP_CH channels: c0p[chunk][k] = sum_c e_c W[c][k], e = a mean invstd q - a m;  per c: dgamma += sum g zhat, dbeta += sum g.
// Thread = k (coalesced rows), a workgroup = PREP_CH channels x 256 k.
constexpr int PREP_CH = 8;
__global__ void __launch_bounds__(256) fold_bwd_prep_kernel(const float *__restrict__ G, const float *__restrict__ P,
                                                            const double *__restrict__ A1, const float *__restrict__ W,
                                                            const double *__restrict__ sums2, const double *__restrict__ sums2_local,
                                                            double count, const float *__restrict__ mean,
                                                            const float *__restrict__ invstd, const float *__restrict__ gamma,
                                                            float *__restrict__ dW, float *__restrict__ dgamma,
                                                            float *__restrict__ dbeta, float *__restrict__ Wd,
                                                            float *__restrict__ c0p, int N, int K) {
    const int k = blockIdx.x * 256 + threadIdx.x, cbase = blockIdx.y * PREP_CH;
    const bool live = k < K;
    const float a1 = live ? (float)A1[k] : 0.f;
    const float *mloc = P + (long)N * K;
    const double inv_count = 1.0 / count;
    float part = 0.f;
#pragma unroll
    for (int cc = 0; cc < PREP_CH; ++cc) {
        const int c = cbase + cc;                            // uniform: the per-channel constants are scalar loads
        if (c >= N) break;
        const float is = invstd[c], mu = mean[c], a = gamma[c] * is;
        const float m = (float)(sums2[c] * inv_count), q = (float)(sums2[N + c] * inv_count);
        const float d = a * is * q, e = d * mu - a * m;
        if (live) {
            const long i = (long)c * K + k;
            const float w = wr(W[i]);
            if (dW) dW[i] += a * (G[i] - m * a1 - q * is * (P[i] - (mu - mloc[c]) * a1));
            Wd[i] = d * w;
            part = fmaf(e, w, part);
        }
        if (k == 0) {
            const double *sl = sums2_local ? sums2_local : sums2;
            if (dgamma) dgamma[c] += (float)sl[N + c];
            if (dbeta) dbeta[c] += (float)sl[c];
        }
    }
    if (live) c0p[(long)blockIdx.y * K + k] = part;
}

// -T = -(sum of the Z partial products of W^T Wd) as the bf16 FORWARD weight operand of the K -> K 1x1 conv dx += x (-T):
// element (row k, kk = j) = -T[j][k] at ((j >> 5) K + k) 32 + (j & 31).  T is symmetric, so thread (a, b) reads T[a][b]
// (coalesced over b) and writes it as row a, kk = b (32 consecutive b = 64 contiguous bytes).  The last K / 32 workgroups add
// the channel chunks' partial sums of c0 instead: 32 columns x 8 interleaved slices each, combined in a fixed order.
__global__ void __launch_bounds__(256) fold_T_finish_kernel(const float *__restrict__ Tpart, int Z, bf16_t *__restrict__ negT, int K,
                                                            const float *__restrict__ c0p, int chunks, float *__restrict__ c0,
                                                            int t_blocks) {
    if ((int)blockIdx.x >= t_blocks) {
        __shared__ float sh[256];
        const int kl = threadIdx.x & 31, sl = threadIdx.x >> 5;
        const int k = ((int)blockIdx.x - t_blocks) * 32 + kl;
        float s = 0.f;
        if (k < K)
            for (int q = sl; q < chunks; q += 8) s += c0p[(long)q * K + k];
        sh[threadIdx.x] = s;
        __syncthreads();
        if (sl == 0 && k < K) {
            float t = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) t += sh[q * 32 + kl];
            c0[k] = t;
        }
        return;
    }
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)K * K) return;
    const int a = (int)(i / K), b = (int)(i - (long)a * K);
    float t = 0.f;
    for (int z = 0; z < Z; ++z) t += Tpart[(long)z * K * K + i];
    negT[((long)(b >> 5) * K + a) * 32 + (b & 31)] = (bf16_t)(-t);
}

}  // namespace

extern "C" size_t rcf_fold_fwd_scratch_bytes(int N, int K) {
    if (N <= 0 || K <= 0 || N % 64 || K % 64) return 0;
    return (size_t)gemm_splits(N, K, K) * N * K * sizeof(float);
}

/* fin == NULL: sums only (SyncBN: all-reduce them, then rcf_fold_finalize_f32); fin != NULL: local statistics, finalized in
 * the same launch (sums may be NULL) */
extern "C" int rcf_fold_fwd_f32(const float *S, const double *A1, const float *W, float *P, double *sums,
                                const rcf_fold_finalize *fin, void *scratch, size_t scratch_bytes, double rows_local, int N, int K,
                                void *stream) {
    if (!S || !A1 || !W || !P || (!sums && !fin) || N <= 0 || K <= 0 || N % 64 || K % 64 || !(rows_local > 0)) return RCF_EINVAL;
    if (fin && fin->count != rows_local) return RCF_EINVAL;          // finalizing here means the statistics ARE the local ones
    if (!scratch || scratch_bytes < rcf_fold_fwd_scratch_bytes(N, K)) return RCF_EWORKSPACE;
    if (fin && (!fin->gamma || !fin->beta || !fin->mean || !fin->invstd || !fin->scale || !fin->shift || !(fin->count > 0)))
        return RCF_EINVAL;
    hipStream_t st = rcf_stream(stream);
    const int Z = gemm_splits(N, K, K);
    hipLaunchKernelGGL((small_gemm_kernel<true, true>), dim3(K / 64, N / 64, Z), dim3(256), 0, st, W, (long)K, 1L, S, (long)K,
                       (float *)scratch, N, K, K / Z, A1, 1.0 / rows_local);
    RCF_LAUNCH_CHECK();
    FoldFin f{};
    if (fin) {
        f.count = fin->count; f.eps = fin->eps; f.momentum = fin->momentum; f.gamma = fin->gamma; f.beta = fin->beta;
        f.mean = fin->mean; f.invstd = fin->invstd; f.scale = fin->scale; f.shift = fin->shift;
        f.rmean = fin->running_mean; f.rvar = fin->running_var; f.nbt = fin->num_batches_tracked; f.on = 1;
    }
    hipLaunchKernelGGL(fold_stats_kernel, dim3(rcf_cdiv(N, 4)), dim3(256), 0, st, (const float *)scratch, Z, A1, W, P, sums, N, K, f,
                       rows_local);
    RCF_LAUNCH_CHECK();
    return 0;
}

extern "C" int rcf_fold_finalize_f32(const double *sums, int C, const rcf_fold_finalize *fin, void *stream) {
    if (!sums || C <= 0 || !fin || !fin->gamma || !fin->beta || !fin->mean || !fin->invstd || !fin->scale || !fin->shift ||
        !(fin->count > 0))
        return RCF_EINVAL;
    FoldFin f{};
    f.count = fin->count; f.eps = fin->eps; f.momentum = fin->momentum; f.gamma = fin->gamma; f.beta = fin->beta;
    f.mean = fin->mean; f.invstd = fin->invstd; f.scale = fin->scale; f.shift = fin->shift;
    f.rmean = fin->running_mean; f.rvar = fin->running_var; f.nbt = fin->num_batches_tracked; f.on = 1;
    hipLaunchKernelGGL(fold_finalize_kernel, dim3(rcf_cdiv(C, 256)), dim3(256), 0, rcf_stream(stream), sums, C, f);
    RCF_LAUNCH_CHECK();
    return 0;
}

namespace {
struct MaskGeom { int chunks; long per; };
inline MaskGeom mask_geom(long rows, int C) {
    const int rpp = 256 / (C >> 3);
    long per = (rows + 2047) / 2048;
    per = (per + rpp - 1) / rpp * rpp;
    if (per < rpp) per = rpp;
    return MaskGeom{(int)((rows + per - 1) / per), per};
}
}  // namespace

extern "C" size_t rcf_relu_mask_colsum_bf16_workspace_bytes(long rows, int C) {
    if (rows <= 0 || C <= 0 || C % 8 || C > 2048 || 256 % (C >> 3)) return 0;
    return (size_t)(mask_geom(rows, C).chunks + 64) * 2 * C * sizeof(double);
}

extern "C" int rcf_relu_mask_colsum_bf16(const void *dy, int dy_pitch, const void *y, int y_pitch, void *g, int g_pitch,
                                         long rows, int C, double *colsums, void *workspace, size_t workspace_bytes,
                                         void *stream) {
    if (!dy || !y || !g || !colsums || rows <= 0 || C <= 0 || C % 8 || C > 2048 || 256 % (C >> 3) || dy_pitch % 8 || y_pitch % 8 || g_pitch % 8 ||
        dy_pitch < C || y_pitch < C || g_pitch < C || !rcf_aligned16(dy) || !rcf_aligned16(y) || !rcf_aligned16(g))
        return RCF_EINVAL;
    if (!workspace || workspace_bytes < rcf_relu_mask_colsum_bf16_workspace_bytes(rows, C)) return RCF_EWORKSPACE;
    const MaskGeom mg = mask_geom(rows, C);
    hipLaunchKernelGGL(relu_mask_colsum_kernel, dim3((unsigned)mg.chunks), dim3(256), 0, rcf_stream(stream), (const bf16_t *)dy,
                       dy_pitch, (const bf16_t *)y, y_pitch, (bf16_t *)g, g_pitch, rows, C, mg.per, (double *)workspace);
    RCF_LAUNCH_CHECK();
    return rcf_sum_partials_bn((const double *)workspace, mg.chunks, C, colsums, (double *)workspace + (size_t)mg.chunks * 2 * C,
                               nullptr, stream);
}

extern "C" int rcf_fold_bwd_sums_f32(const float *G, const float *W, const double *colsums, const float *mean,
                                     const float *invstd, double *sums2, int N, int K, void *stream) {
    if (!G || !W || !colsums || !mean || !invstd || !sums2 || N <= 0 || K <= 0) return RCF_EINVAL;
    hipLaunchKernelGGL(fold_bwd_sums_kernel, dim3(rcf_cdiv(N, 4)), dim3(256), 0, rcf_stream(stream), G, W, colsums, mean, invstd,
                       sums2, N, K);
    RCF_LAUNCH_CHECK();
    return 0;
}

extern "C" size_t rcf_fold_bwd_scratch_bytes(int N, int K) {
    if (N <= 0 || K <= 0 || N % 64 || K % 64) return 0;
    return ((size_t)N * K + (size_t)rcf_cdiv(N, PREP_CH) * K + (size_t)gemm_splits(K, K, N) * K * K) * sizeof(float);
}

extern "C" int rcf_fold_wg_bf16(const float *W, const float *scale, void *wg_t_bf16, int N, int K, void *stream) {
    if (!W || !scale || !wg_t_bf16 || N <= 0 || K <= 0 || N % 32 || K % 8) return RCF_EINVAL;
    hipLaunchKernelGGL(fold_wg_kernel, dim3(rcf_cdiv((long)N * K, 256)), dim3(256), 0, rcf_stream(stream), W, scale,
                       (bf16_t *)wg_t_bf16, N, K);
    RCF_LAUNCH_CHECK();
    return 0;
}

extern "C" int rcf_fold_bwd_prepare_f32(const float *G, const float *P, const double *A1, const float *W, const double *sums2,
                                        const double *sums2_local, double count, const float *mean, const float *invstd,
                                        const float *gamma, float *dW, float *dgamma, float *dbeta, void *negT_bf16, float *c0,
                                        void *scratch, size_t scratch_bytes, int N, int K, void *stream) {
    if (!G || !P || !A1 || !W || !sums2 || !mean || !invstd || !gamma || !negT_bf16 || !c0 || N <= 0 || K <= 0 || N % 64 || K % 64 ||
        !(count > 0))
        return RCF_EINVAL;
    if (!scratch || scratch_bytes < rcf_fold_bwd_scratch_bytes(N, K)) return RCF_EWORKSPACE;
    hipStream_t st = rcf_stream(stream);
    const int chunks = rcf_cdiv(N, PREP_CH);
    float *Wd = (float *)scratch, *c0p = Wd + (size_t)N * K, *Tpart = c0p + (size_t)chunks * K;
    hipLaunchKernelGGL(fold_bwd_prep_kernel, dim3(rcf_cdiv(K, 256), chunks), dim3(256), 0, st, G, P, A1, W, sums2, sums2_local, count,
                       mean, invstd, gamma, dW, dgamma, dbeta, Wd, c0p, N, K);
    RCF_LAUNCH_CHECK();
    // T = W^T Wd: A(j, c) = W[c][j] (contiguous along j), B(c, k) = Wd[c][k]; Z splits of the channels, summed by the finish kernel
    const int Z = gemm_splits(K, K, N);
    hipLaunchKernelGGL(small_gemm_kernel<true>, dim3(K / 64, K / 64, Z), dim3(256), 0, st, W, 1L, (long)K, (const float *)Wd, (long)K,
                       Tpart, K, K, N / Z);
    RCF_LAUNCH_CHECK();
    const int t_blocks = (int)rcf_cdiv((long)K * K, 256);
    hipLaunchKernelGGL(fold_T_finish_kernel, dim3(t_blocks + rcf_cdiv(K, 32)), dim3(256), 0, st, (const float *)Tpart, Z,
                       (bf16_t *)negT_bf16, K, (const float *)c0p, chunks, c0, t_blocks);
    RCF_LAUNCH_CHECK();
    return 0;
}
