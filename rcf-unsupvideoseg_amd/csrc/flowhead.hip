// Flow-aggregation head + loss tail of RCF's training step, forward and backward.
// Reference: models/flow_aggregation_head_with_residual.py:235-310 (aggregate_flow_with_residual),
// :164-233 (get_demean_affine_flow: per-segment weighted least squares), :150-162 (clamp), :312-399
// (forward, L1 / robust loss); models/rcf_model.py:433-434 (softmax, log_softmax of the softmax),
// :376-378 (entropy), :350-374 (sharpen: KL to p^(1/T) or object-aware hinge), :380-408 (asymmetric
// clamped MSE against pl / crf targets); models/compactness_head.py:14-57 (compactness of one channel).
//
// The reference materialises [B,64,C,h,w] and [B,C,hw,2,2] intermediates and solves the 2x2 (5x5)
// systems through a batched LU library call.  Here every stage is one pass over the pixels of the
// 2B "direction images" (fw uses frame 0's masks, bw frame 1's):
//   softmax        logits -> p (planar), S_c = sum_px p_c, entropy / target losses
//   pool           g[k][c] = sum_px f[px][k] p_c(px) / S_c      lanes = the 64 flow features (coalesced rows)
//   mlp            64->64 LeakyReLU ->2 per segment; affine: fp64 moments -> A_c = S_Fw S_ww^-1 (Gaussian
//                  elimination with pivoting in fp64, in-kernel)
//   recon          overall = sum_c u_c p_c (+ sum_c p_c A_c (w - mu_c)) + scale * sum_c tanh(r_c/div) p_c,
//                  loss = mean |gt - overall| (or (|.|+eps)^q); G = d loss / d overall kept for backward
// and the analytic backward (derivation in DESIGN.md §4.5) in four passes.  All sums that feed a
// division or a subtraction accumulate in fp64 (wavefront shuffle reduction -> per-block partials ->
// fixed-order final sum: deterministic, no float atomics).  HBM-bound: ~100 B per pixel per direction.
#include "rcf_common.h"

namespace {

constexpr int NF = 64;        // flow feature channels (num_flow_feat_channels)
constexpr int CMAX = 8;       // max segments
constexpr int DMAX = 5;       // affine basis size: 2 (row, col) or 5 (+ row^2, col^2, row*col)
constexpr int RB = 256;       // threads per block for pixel passes
constexpr int NCHUNK = 64;    // pixel chunks per image for reductions

struct Ws {                   // workspace carve-up (all device pointers)
    float *p;                 // [NB][C][P]
    float *gt;                // [NB][2][P] clamped flow
    float *G;                 // [NB][2][P] d loss / d overall
    float *gw;                // [NB][C][P] d loss / d w_c(px)
    float *g;                 // [NB][NF][C] pooled features
    float *u1;                // [NB][NF][C] pre-activation of the first 1x1
    float *u;                 // [NB][2][C]
    float *dg;                // [NB][NF][C]
    double *S;                // [NB][C]
    double *mu;               // [NB][C][2 + D]   mu_F (2), mu_w (D)
    double *A;                // [NB][C][2][D]
    double *Qinv;             // [NB][C][D][D]
    double *GM;               // [NB][C][2][D]
    double *GQ;               // [NB][C][D][D]
    double *Gmu;              // [NB][C][D]
    double *T;                // [NB][C] sum_px w gw
    double *part;             // [NB][NCHUNK][PARTW] block partials
    double *red;              // [NB][PARTW] reduced partials
    double *loss;             // [8]: seg_fw, seg_bw, entropy, target0, target1, compactness, sharpen
    double *cen;              // [NB][2] soft centroid (row/h, col/w) of the compact channel
    float *pgrad;             // [NB][NF*NF + NF + 2*NF + 2] per-image MLP parameter gradients
    float *poolpart;          // [NB][NCHUNK][NF][CMAX] pooling partials
};
constexpr int PARTW = 96;     // max doubles per partial row

struct Cfg {
    int B, NB, C, P, h, w, Cp, D, robust, tanh_res;
    float eps, q, clamp_t, res_scale, div_coeff, w_seg, w_entropy;
    int ntgt, tgt_channel;
    float t_wpos[2], t_wneg[2], t_w[2], t_th[2];
    float w_compact;
    int comp_ch;
    float w_sharpen, t_sharpen;
    int sharpen_mode;
};

inline size_t al(size_t x) { return (x + 255) & ~(size_t)255; }
size_t carve(char *base, const Cfg &c, Ws &w) {
    size_t o = 0;
    auto take = [&](size_t bytes) { char *p = base ? base + o : nullptr; o += al(bytes); return p; };
    const size_t NB = c.NB, C = c.C, P = c.P, D = c.D > 0 ? c.D : 1;
    w.p = (float *)take(NB * C * P * 4);
    w.gt = (float *)take(NB * 2 * P * 4);
    w.G = (float *)take(NB * 2 * P * 4);
    w.gw = (float *)take(NB * C * P * 4);
    w.g = (float *)take(NB * NF * C * 4);
    w.u1 = (float *)take(NB * NF * C * 4);
    w.u = (float *)take(NB * 2 * C * 4);
    w.dg = (float *)take(NB * NF * C * 4);
    w.S = (double *)take(NB * C * 8);
    w.mu = (double *)take(NB * C * (2 + D) * 8);
    w.A = (double *)take(NB * C * 2 * D * 8);
    w.Qinv = (double *)take(NB * C * D * D * 8);
    w.GM = (double *)take(NB * C * 2 * D * 8);
    w.GQ = (double *)take(NB * C * D * D * 8);
    w.Gmu = (double *)take(NB * C * D * 8);
    w.T = (double *)take(NB * C * 8);
    w.part = (double *)take(NB * NCHUNK * PARTW * 8);
    w.red = (double *)take(NB * PARTW * 8);
    w.loss = (double *)take(8 * 8);
    w.cen = (double *)take(NB * 2 * 8);
    w.pgrad = (float *)take(NB * (NF * NF + NF + 2 * NF + 2) * 4);
    w.poolpart = (float *)take(NB * NCHUNK * NF * CMAX * 4);
    return o;
}

__device__ __forceinline__ void basis(int px, int w, int D, float *om) {
    const int y = px / w, x = px - y * w;
    om[0] = (float)y;
    om[1] = (float)x;
    if (D > 2) { om[2] = (float)(y * y); om[3] = (float)(x * x); om[4] = (float)(y * x); }
}

// block-level: reduce `nv` doubles held per thread (v[]) across the block, thread 0 writes dst[0..nv)
template <int NV>
__device__ __forceinline__ void block_reduce_store(double (&v)[NV], int nv, double *dst) {
    __shared__ double sh[4 * NV];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NV; i++) {
        if (i < nv) {
            const double s = wave_sum_d(v[i]);
            if (lane == 0) sh[wv * NV + i] = s;
        }
    }
    __syncthreads();
    if (threadIdx.x < nv) dst[threadIdx.x] = sh[threadIdx.x] + sh[NV + threadIdx.x] + sh[2 * NV + threadIdx.x] + sh[3 * NV + threadIdx.x];
    __syncthreads();
}

// red[n][i] = sum_chunk part[n][chunk][i]
__global__ void reduce_parts_kernel(const double *__restrict__ part, double *__restrict__ red, int nv) {
    const int n = blockIdx.x, i = threadIdx.x;
    if (i >= nv) return;
    double s = 0;
    for (int k = 0; k < NCHUNK; k++) s += part[((long)n * NCHUNK + k) * PARTW + i];
    red[(long)n * PARTW + i] = s;
}

// ------------------------------------------------------------------------------- prepare
// gfw/gbw [B][2][P] -> gt [NB][2][P] (clamped) and flow4 [NB][P][4] (NHWC input of the first conv)
__global__ void __launch_bounds__(RB) prepare_kernel(Cfg c, const float *__restrict__ gfw, const float *__restrict__ gbw,
                                                     float *__restrict__ gt, float *__restrict__ flow4) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)c.NB * c.P) return;
    const int n = (int)(i / c.P), px = (int)(i - (long)n * c.P);
    const int b = n >> 1, d = n & 1;
    const float *src = (d ? gbw : gfw) + (long)b * 2 * c.P;
    float fx = src[px], fy = src[c.P + px];
    if (c.clamp_t >= 0.f) {
        fx = fminf(fmaxf(fx, -c.clamp_t), c.clamp_t);
        fy = fminf(fmaxf(fy, -c.clamp_t), c.clamp_t);
    }
    gt[((long)n * 2) * c.P + px] = fx;
    gt[((long)n * 2 + 1) * c.P + px] = fy;
    *reinterpret_cast<f32x4 *>(flow4 + i * 4) = f32x4{fx, fy, 0.f, 0.f};
}

// ------------------------------------------------------------------------------- softmax + scalar losses
// partial row: [0..C) sum p_c ; [C] entropy sum ; [C+1], [C+2] target loss sums ; [C+3], [C+4] sum (row/h) m,
// sum (col/w) m of the compact channel ; [C+5] sharpen loss sum
__global__ void __launch_bounds__(RB) softmax_kernel(Cfg c, Ws w, const float *__restrict__ logits,
                                                     const float *__restrict__ tgt0, const float *__restrict__ tgt1) {
    const int n = blockIdx.y, chunk = blockIdx.x;
    const int per = (c.P + NCHUNK - 1) / NCHUNK;
    const int p0 = chunk * per, p1 = min(c.P, p0 + per);
    double acc[CMAX + 6];
    for (int i = 0; i < CMAX + 6; i++) acc[i] = 0;
    for (int px = p0 + threadIdx.x; px < p1; px += RB) {
        const float *l = logits + ((long)n * c.P + px) * c.Cp;
        float v[CMAX], mx = -INFINITY;
        for (int k = 0; k < c.C; k++) { v[k] = l[k]; mx = fmaxf(mx, v[k]); }
        float s = 0.f;
        for (int k = 0; k < c.C; k++) { v[k] = expf(v[k] - mx); s += v[k]; }
        float se = 0.f;
        for (int k = 0; k < c.C; k++) { v[k] = v[k] / s; w.p[((long)n * c.C + k) * c.P + px] = v[k]; acc[k] += (double)v[k]; }
        // log_softmax of the probabilities (models/rcf_model.py:434), entropy -(p * logp).sum
        float pm = v[0];
        for (int k = 1; k < c.C; k++) pm = fmaxf(pm, v[k]);
        for (int k = 0; k < c.C; k++) se += expf(v[k] - pm);
        const float lse = pm + logf(se);
        float ent = 0.f;
        for (int k = 0; k < c.C; k++) ent -= v[k] * (v[k] - lse);
        acc[c.C] += (double)ent;
        for (int t = 0; t < c.ntgt; t++) {
            float tv = (t ? tgt1 : tgt0)[(long)n * c.P + px];
            if (c.t_th[t] != -1.f) tv = tv > c.t_th[t] ? 1.f : 0.f;
            const float d = tv - v[c.tgt_channel];
            acc[c.C + 1 + t] += (double)(d > 0.f ? c.t_wpos[t] * d * d : c.t_wneg[t] * d * d);
        }
        if (c.w_compact != 0.f) {
            const int y = px / c.w, x = px - y * c.w;
            const float m = v[c.comp_ch];
            acc[c.C + 3] += (double)(((float)y / (float)c.h) * m);
            acc[c.C + 4] += (double)(((float)x / (float)c.w) * m);
        }
        if (c.sharpen_mode == 1) {
            // F.kl_div(log_softmax(p), sharpen(p, T)) summed over channels: xlogy(t, t) - t * logp
            float tt[CMAX], ts = 0.f;
            for (int k = 0; k < c.C; k++) { tt[k] = powf(v[k], 1.f / c.t_sharpen); ts += tt[k]; }
            float kl = 0.f;
            for (int k = 0; k < c.C; k++) {
                const float t = tt[k] / ts;
                kl += (t > 0.f ? t * logf(t) : 0.f) - t * (v[k] - lse);
            }
            acc[c.C + 5] += (double)kl;
        } else if (c.sharpen_mode == 2) {
            float mo = 0.f;                                // max over the other channels (object channel zeroed)
            for (int k = 0; k < c.C; k++) if (k != c.tgt_channel) mo = fmaxf(mo, v[k]);
            acc[c.C + 5] += (double)fmaxf(c.t_sharpen - fabsf(v[c.tgt_channel] - mo), 0.f);
        }
    }
    block_reduce_store<CMAX + 6>(acc, c.C + 6, w.part + ((long)n * NCHUNK + chunk) * PARTW);
}

__global__ void softmax_final_kernel(Cfg c, Ws w) {
    // one block; S[n][c] and the scalar losses
    const int t = threadIdx.x;
    // A segment whose softmax mass is EXACTLY zero in a frame (logits hundreds apart: exp underflows on every pixel) makes
    // the reference divide 0 by 0 (mask / mask.sum, flow_aggregation_head_with_residual.py:242-243) and the whole step NaN.
    // Here such a segment is absent from the frame: its mass is stored as +inf, so every x / S below is 0 -- its pooled flow,
    // its weights p / S and the gradient through them.  Any non-zero mass is used as it is (results unchanged).
    if (t < c.NB * c.C) {
        const double m = w.red[(long)(t / c.C) * PARTW + (t % c.C)];
        w.S[t] = m > 0.0 ? m : (double)INFINITY;
    }
    if (c.w_compact != 0.f && t < c.NB * 2)
        w.cen[t] = w.red[(long)(t >> 1) * PARTW + c.C + 3 + (t & 1)] / w.red[(long)(t >> 1) * PARTW + c.comp_ch];
    if (t == 0) {
        double ent = 0, t0 = 0, t1 = 0, sh = 0;
        for (int n = 0; n < c.NB; n++) {
            ent += w.red[(long)n * PARTW + c.C]; t0 += w.red[(long)n * PARTW + c.C + 1]; t1 += w.red[(long)n * PARTW + c.C + 2];
            sh += w.red[(long)n * PARTW + c.C + 5];
        }
        const double cnt = (double)c.NB * c.P;
        w.loss[2] = ent / cnt;
        w.loss[3] = t0 / cnt;
        w.loss[4] = t1 / cnt;
        w.loss[6] = c.sharpen_mode == 1 ? sh / (cnt * c.C) : sh / cnt;
        w.loss[7] = 0;
    }
}

// ------------------------------------------------------------------------------- pooling
// partial[n][chunk][k*C + c] = sum_{px in chunk} f[px][k] p_c(px);  lanes = features
__global__ void __launch_bounds__(RB) pool_kernel(Cfg c, Ws w, const float *__restrict__ feat) {
    const int n = blockIdx.y, chunk = blockIdx.x;
    const int per = (c.P + NCHUNK - 1) / NCHUNK;
    const int p0 = chunk * per, p1 = min(c.P, p0 + per);
    const int k = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float acc[CMAX];
    for (int i = 0; i < CMAX; i++) acc[i] = 0.f;
    for (int px = p0 + wv; px < p1; px += 4) {
        const float f = feat[((long)n * c.P + px) * NF + k];
        for (int cc = 0; cc < c.C; cc++) acc[cc] += f * w.p[((long)n * c.C + cc) * c.P + px];
    }
    __shared__ float sh[4][NF][CMAX];
    for (int cc = 0; cc < c.C; cc++) sh[wv][k][cc] = acc[cc];
    __syncthreads();
    if (wv == 0) {
        float *dst = w.poolpart + ((long)n * NCHUNK + chunk) * NF * CMAX;
        for (int cc = 0; cc < c.C; cc++) dst[k * CMAX + cc] = sh[0][k][cc] + sh[1][k][cc] + sh[2][k][cc] + sh[3][k][cc];
    }
}

__global__ void pool_final_kernel(Cfg c, Ws w) {
    const int n = blockIdx.x, k = threadIdx.x & 63, cc = threadIdx.x >> 6;
    if (cc >= c.C) return;
    double s = 0;
    for (int ch = 0; ch < NCHUNK; ch++) s += (double)w.poolpart[((long)n * NCHUNK + ch) * NF * CMAX + k * CMAX + cc];
    w.g[((long)n * NF + k) * c.C + cc] = (float)(s / w.S[(long)n * c.C + cc]);
}

// ------------------------------------------------------------------------------- MLP (Conv1d k=1 x2)
// W1 [NF][NF], b1 [NF], W2 [2][NF], b2 [2]
__global__ void __launch_bounds__(256) mlp_kernel(Cfg c, Ws w, const float *__restrict__ W1, const float *__restrict__ b1,
                                                  const float *__restrict__ W2, const float *__restrict__ b2) {
    const int n = blockIdx.x, j = threadIdx.x & 63, cc = threadIdx.x >> 6;
    __shared__ float act[NF][CMAX];
    for (int c0 = cc; c0 < c.C; c0 += 4) {
        float s = b1[j];
        for (int k = 0; k < NF; k++) s += W1[j * NF + k] * w.g[((long)n * NF + k) * c.C + c0];
        w.u1[((long)n * NF + j) * c.C + c0] = s;
        act[j][c0] = s > 0.f ? s : 0.1f * s;
    }
    __syncthreads();
    if (threadIdx.x < 2 * c.C) {
        const int dd = threadIdx.x / c.C, c0 = threadIdx.x % c.C;
        float s = b2[dd];
        for (int k = 0; k < NF; k++) s += W2[dd * NF + k] * act[k][c0];
        w.u[((long)n * 2 + dd) * c.C + c0] = s;
    }
}

// ------------------------------------------------------------------------------- affine least squares
// pass 1 partial row per c: sum p F (2), sum p w (D)        -> mu = ./S
__global__ void __launch_bounds__(RB) moments1_kernel(Cfg c, Ws w) {
    const int n = blockIdx.y, chunk = blockIdx.x;
    const int per = (c.P + NCHUNK - 1) / NCHUNK;
    const int p0 = chunk * per, p1 = min(c.P, p0 + per);
    const int nv = 2 + c.D;
    for (int cc = 0; cc < c.C; cc++) {
        double acc[2 + DMAX];
        for (int i = 0; i < 2 + DMAX; i++) acc[i] = 0;
        for (int px = p0 + threadIdx.x; px < p1; px += RB) {
            const double pv = w.p[((long)n * c.C + cc) * c.P + px];
            float om[DMAX];
            basis(px, c.w, c.D, om);
            acc[0] += pv * w.gt[((long)n * 2) * c.P + px];
            acc[1] += pv * w.gt[((long)n * 2 + 1) * c.P + px];
            for (int j = 0; j < c.D; j++) acc[2 + j] += pv * om[j];
        }
        block_reduce_store<2 + DMAX>(acc, nv, w.part + ((long)n * NCHUNK + chunk) * PARTW + cc * nv);
    }
}
__global__ void moments1_final_kernel(Cfg c, Ws w) {
    const int n = blockIdx.x, t = threadIdx.x, nv = 2 + c.D;
    if (t >= c.C * nv) return;
    w.mu[(long)n * c.C * nv + t] = w.red[(long)n * PARTW + t] / w.S[(long)n * c.C + t / nv];
}
// pass 2 partial row per c: sum p (F-muF)(w-muw)^T (2D), sum p (w-muw)(w-muw)^T upper triangle (D(D+1)/2)
__global__ void __launch_bounds__(RB) moments2_kernel(Cfg c, Ws w, int cc) {
    const int n = blockIdx.y, chunk = blockIdx.x;
    const int per = (c.P + NCHUNK - 1) / NCHUNK;
    const int p0 = chunk * per, p1 = min(c.P, p0 + per);
    const int D = c.D, nv = 2 * D + D * (D + 1) / 2;
    const double *mu = w.mu + ((long)n * c.C + cc) * (2 + D);
    double acc[2 * DMAX + DMAX * (DMAX + 1) / 2];
    for (int i = 0; i < 2 * DMAX + DMAX * (DMAX + 1) / 2; i++) acc[i] = 0;
    for (int px = p0 + threadIdx.x; px < p1; px += RB) {
        const double pv = w.p[((long)n * c.C + cc) * c.P + px];
        float om[DMAX];
        basis(px, c.w, D, om);
        const double f0 = (double)w.gt[((long)n * 2) * c.P + px] - mu[0], f1 = (double)w.gt[((long)n * 2 + 1) * c.P + px] - mu[1];
        double od[DMAX];
        for (int j = 0; j < D; j++) od[j] = (double)om[j] - mu[2 + j];
        int q = 2 * D;
        for (int j = 0; j < D; j++) {
            acc[j] += pv * f0 * od[j];
            acc[D + j] += pv * f1 * od[j];
            for (int l = j; l < D; l++) acc[q++] += pv * od[j] * od[l];
        }
    }
    block_reduce_store<2 * DMAX + DMAX * (DMAX + 1) / 2>(acc, nv, w.part + ((long)n * NCHUNK + chunk) * PARTW);
}
// A = M Q^-1 and Q^-1 by Gauss-Jordan with partial pivoting (fp64); one thread per (n, c)
__global__ void affine_solve_kernel(Cfg c, Ws w, int cc) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= c.NB) return;
    const int D = c.D;
    const double S = w.S[(long)n * c.C + cc];
    const double *r = w.red + (long)n * PARTW;
    if (isinf(S)) {                                        // segment absent from this frame (see softmax_final_kernel)
        double *Qi0 = w.Qinv + ((long)n * c.C + cc) * D * D, *A0 = w.A + ((long)n * c.C + cc) * 2 * D;
        for (int j = 0; j < D * D; j++) Qi0[j] = 0.0;
        for (int j = 0; j < 2 * D; j++) A0[j] = 0.0;
        return;
    }
    double M[2][DMAX], Q[DMAX][2 * DMAX];
    int q = 2 * D;
    for (int j = 0; j < D; j++) {
        M[0][j] = r[j] / S;
        M[1][j] = r[D + j] / S;
        for (int l = j; l < D; l++) { const double v = r[q++] / S; Q[j][l] = v; Q[l][j] = v; }
    }
    for (int j = 0; j < D; j++)
        for (int l = 0; l < D; l++) Q[j][D + l] = (j == l) ? 1.0 : 0.0;
    for (int col = 0; col < D; col++) {
        int piv = col;
        for (int rr = col + 1; rr < D; rr++) if (fabs(Q[rr][col]) > fabs(Q[piv][col])) piv = rr;
        if (piv != col) for (int l = 0; l < 2 * D; l++) { const double t = Q[col][l]; Q[col][l] = Q[piv][l]; Q[piv][l] = t; }
        const double inv = 1.0 / Q[col][col];
        for (int l = 0; l < 2 * D; l++) Q[col][l] *= inv;
        for (int rr = 0; rr < D; rr++) {
            if (rr == col) continue;
            const double fct = Q[rr][col];
            for (int l = 0; l < 2 * D; l++) Q[rr][l] -= fct * Q[col][l];
        }
    }
    double *Qi = w.Qinv + ((long)n * c.C + cc) * D * D, *A = w.A + ((long)n * c.C + cc) * 2 * D;
    for (int j = 0; j < D; j++)
        for (int l = 0; l < D; l++) Qi[j * D + l] = Q[j][D + l];
    for (int dd = 0; dd < 2; dd++)
        for (int l = 0; l < D; l++) {
            double s = 0;
            for (int j = 0; j < D; j++) s += M[dd][j] * Q[j][D + l];
            A[dd * D + l] = s;
        }
}

// ------------------------------------------------------------------------------- reconstruction + loss
// R: resized residual NHWC [B][P][4C] (channel d*2C + dd*C + c).  Optional flow planes [NB][2][P].
__global__ void __launch_bounds__(RB) recon_kernel(Cfg c, Ws w, const float *__restrict__ R, float *__restrict__ o_pred,
                                                   float *__restrict__ o_agg, float *__restrict__ o_adj,
                                                   float *__restrict__ o_aff) {
    const int n = blockIdx.y, chunk = blockIdx.x;
    const int per = (c.P + NCHUNK - 1) / NCHUNK;
    const int p0 = chunk * per, p1 = min(c.P, p0 + per);
    const int b = n >> 1, d = n & 1, D = c.D;
    const float gcoef = c.w_seg / ((float)c.B * 2.f * (float)c.P);
    double lacc[2] = {0, 0};
    const float ycen = c.w_compact != 0.f ? (float)w.cen[2 * n] : 0.f, xcen = c.w_compact != 0.f ? (float)w.cen[2 * n + 1] : 0.f;
    for (int px = p0 + threadIdx.x; px < p1; px += RB) {
        float pv[CMAX];
        for (int cc = 0; cc < c.C; cc++) pv[cc] = w.p[((long)n * c.C + cc) * c.P + px];
        float om[DMAX];
        if (D) basis(px, c.w, D, om);
        if (c.w_compact != 0.f) {
            const int y = px / c.w, x = px - y * c.w;
            const float dy = (float)y / (float)c.h - ycen, dx = (float)x / (float)c.w - xcen;
            lacc[1] += (double)((dy * dy + dx * dx) * pv[c.comp_ch]);
        }
        const float *r = R + ((long)b * c.P + px) * 4 * c.C + d * 2 * c.C;
        for (int dd = 0; dd < 2; dd++) {
            float agg = 0.f, adj = 0.f, aff = 0.f;
            for (int cc = 0; cc < c.C; cc++) {
                agg += w.u[((long)n * 2 + dd) * c.C + cc] * pv[cc];
                const float rv = r[dd * c.C + cc];
                adj += (c.tanh_res ? tanhf(rv / c.div_coeff) : rv) * pv[cc];
                if (D) {
                    const double *A = w.A + ((long)n * c.C + cc) * 2 * D + dd * D;
                    const double *mu = w.mu + ((long)n * c.C + cc) * (2 + D) + 2;
                    float s = 0.f;
                    for (int j = 0; j < D; j++) s += (float)A[j] * (om[j] - (float)mu[j]);
                    aff += pv[cc] * s;
                }
            }
            if (c.tanh_res) adj *= c.res_scale;
            const float overall = D ? (agg + aff + adj) : (agg + adj);
            const float diff = w.gt[((long)n * 2 + dd) * c.P + px] - overall;
            const float ad = fabsf(diff);
            float lv, gv;                       // loss value, d loss / d overall
            const float sgn = diff > 0.f ? -1.f : (diff < 0.f ? 1.f : 0.f);
            if (c.robust) { lv = powf(ad + c.eps, c.q); gv = sgn * c.q * powf(ad + c.eps, c.q - 1.f); }
            else { lv = ad; gv = sgn; }
            lacc[0] += (double)lv;
            w.G[((long)n * 2 + dd) * c.P + px] = gv * gcoef;
            const long oi = ((long)n * 2 + dd) * c.P + px;
            if (o_pred) o_pred[oi] = overall;
            if (o_agg) o_agg[oi] = agg;
            if (o_adj) o_adj[oi] = adj;
            if (o_aff && D) o_aff[oi] = aff;
        }
    }
    block_reduce_store<2>(lacc, 2, w.part + ((long)n * NCHUNK + chunk) * PARTW);
}
__global__ void recon_final_kernel(Cfg c, Ws w) {
    if (threadIdx.x != 0) return;
    double fw = 0, bw = 0, comp = 0;
    for (int n = 0; n < c.NB; n++) { ((n & 1) ? bw : fw) += w.red[(long)n * PARTW]; comp += w.red[(long)n * PARTW + 1]; }
    const double cnt = (double)c.B * 2 * c.P;
    w.loss[0] = fw / cnt;
    w.loss[1] = bw / cnt;
    w.loss[5] = comp / ((double)c.NB * c.P);
}

// ------------------------------------------------------------------------------- backward pass 1
// partial row: GP[dd][c] (2C) then per c: GA[c][dd][j] (2D each)
__global__ void __launch_bounds__(RB) bwd_reduce_kernel(Cfg c, Ws w) {
    const int n = blockIdx.y, chunk = blockIdx.x;
    const int per = (c.P + NCHUNK - 1) / NCHUNK;
    const int p0 = chunk * per, p1 = min(c.P, p0 + per);
    const int D = c.D;
    for (int cc = 0; cc < c.C; cc++) {
        double acc[2 + 2 * DMAX];
        for (int i = 0; i < 2 + 2 * DMAX; i++) acc[i] = 0;
        const double *mu = w.mu + ((long)n * c.C + cc) * (2 + D) + 2;
        for (int px = p0 + threadIdx.x; px < p1; px += RB) {
            const double pv = w.p[((long)n * c.C + cc) * c.P + px];
            const double g0 = w.G[((long)n * 2) * c.P + px], g1 = w.G[((long)n * 2 + 1) * c.P + px];
            acc[0] += g0 * pv;
            acc[1] += g1 * pv;
            if (D) {
                float om[DMAX];
                basis(px, c.w, D, om);
                for (int j = 0; j < D; j++) {
                    const double od = (double)om[j] - mu[j];
                    acc[2 + j] += g0 * pv * od;
                    acc[2 + D + j] += g1 * pv * od;
                }
            }
        }
        block_reduce_store<2 + 2 * DMAX>(acc, 2 + 2 * D, w.part + ((long)n * NCHUNK + chunk) * PARTW + cc * (2 + 2 * D));
    }
}

// MLP backward + affine backward, one block per direction image.  Per-image parameter gradients go to
// pgrad[n] = [dW1 (NF*NF) | db1 (NF) | dW2 (2*NF) | db2 (2)]
__global__ void __launch_bounds__(256) mlp_bwd_kernel(Cfg c, Ws w, const float *__restrict__ W1, const float *__restrict__ W2) {
    const int n = blockIdx.x, t = threadIdx.x, D = c.D, C = c.C;
    const int rw = 2 + 2 * D;
    __shared__ float du[2][CMAX], d1[NF][CMAX], gs[NF][CMAX], act[NF][CMAX];
    if (t < 2 * C) du[t / C][t % C] = (float)w.red[(long)n * PARTW + (t % C) * rw + (t / C)];
    for (int i = t; i < NF * C; i += 256) {
        const int j = i / C, cc = i % C;
        const float pre = w.u1[((long)n * NF + j) * C + cc];
        act[j][cc] = pre > 0.f ? pre : 0.1f * pre;
        gs[j][cc] = w.g[((long)n * NF + j) * C + cc];
    }
    __syncthreads();
    float *pg = w.pgrad + (long)n * (NF * NF + NF + 2 * NF + 2);
    for (int i = t; i < NF * C; i += 256) {
        const int j = i / C, cc = i % C;
        const float pre = w.u1[((long)n * NF + j) * C + cc];
        const float da = W2[j] * du[0][cc] + W2[NF + j] * du[1][cc];
        d1[j][cc] = da * (pre > 0.f ? 1.f : 0.1f);
    }
    __syncthreads();
    for (int i = t; i < NF * NF; i += 256) {       // dW1[j][k] = sum_c d1[j][c] g[k][c]
        const int j = i / NF, k = i % NF;
        float s = 0.f;
        for (int cc = 0; cc < C; cc++) s += d1[j][cc] * gs[k][cc];
        pg[i] = s;
    }
    if (t < NF) {
        float s = 0.f;
        for (int cc = 0; cc < C; cc++) s += d1[t][cc];
        pg[NF * NF + t] = s;
    }
    if (t < 2 * NF) {                               // dW2[dd][j] = sum_c du[dd][c] act[j][c]
        const int dd = t / NF, j = t % NF;
        float s = 0.f;
        for (int cc = 0; cc < C; cc++) s += du[dd][cc] * act[j][cc];
        pg[NF * NF + NF + t] = s;
    }
    if (t < 2) {
        float s = 0.f;
        for (int cc = 0; cc < C; cc++) s += du[t][cc];
        pg[NF * NF + NF + 2 * NF + t] = s;
    }
    for (int i = t; i < NF * C; i += 256) {         // dg[k][c] = sum_j W1[j][k] d1[j][c]
        const int k = i / C, cc = i % C;
        float s = 0.f;
        for (int j = 0; j < NF; j++) s += W1[j * NF + k] * d1[j][cc];
        w.dg[((long)n * NF + k) * C + cc] = s;
    }
    if (D && t < C) {                               // affine: GM = GA Q^-1, GQ = -A^T GA Q^-1, Gmu = -A^T GP
        const int cc = t;
        const double *Qi = w.Qinv + ((long)n * C + cc) * D * D, *A = w.A + ((long)n * C + cc) * 2 * D;
        const double *r = w.red + (long)n * PARTW + cc * rw;
        double GA[2][DMAX], GM[2][DMAX];
        for (int j = 0; j < D; j++) { GA[0][j] = r[2 + j]; GA[1][j] = r[2 + D + j]; }
        for (int dd = 0; dd < 2; dd++)
            for (int l = 0; l < D; l++) {
                double s = 0;
                for (int j = 0; j < D; j++) s += GA[dd][j] * Qi[j * D + l];
                GM[dd][l] = s;
                w.GM[((long)n * C + cc) * 2 * D + dd * D + l] = s;
            }
        for (int j = 0; j < D; j++) {
            for (int l = 0; l < D; l++)
                w.GQ[((long)n * C + cc) * D * D + j * D + l] = -(A[j] * GM[0][l] + A[D + j] * GM[1][l]);
            w.Gmu[((long)n * C + cc) * D + j] = -(A[j] * r[0] + A[D + j] * r[1]);
        }
    }
}

// pgrad[n] summed over n in a fixed order and added to the parameter gradients
__global__ void pgrad_final_kernel(Cfg c, Ws w, float *__restrict__ dW1, float *__restrict__ db1, float *__restrict__ dW2,
                                   float *__restrict__ db2) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int tot = NF * NF + NF + 2 * NF + 2;
    if (i >= tot) return;
    float s = 0.f;
    for (int n = 0; n < c.NB; n++) s += w.pgrad[(long)n * tot + i];
    if (i < NF * NF) dW1[i] += s;
    else if (i < NF * NF + NF) db1[i - NF * NF] += s;
    else if (i < NF * NF + 3 * NF) dW2[i - NF * NF - NF] += s;
    else db2[i - NF * NF - 3 * NF] += s;
}

// ------------------------------------------------------------------------------- backward pass 2
// one wavefront per pixel: lanes = features.  gw_c(px) = sum_k dg[k][c] f[px][k] (+ affine terms);
// partial T_c = sum_px w_c gw_c
__global__ void __launch_bounds__(RB) bwd_gw_kernel(Cfg c, Ws w, const float *__restrict__ feat) {
    const int n = blockIdx.y, chunk = blockIdx.x;
    const int per = (c.P + NCHUNK - 1) / NCHUNK;
    const int p0 = chunk * per, p1 = min(c.P, p0 + per);
    const int k = threadIdx.x & 63, wv = threadIdx.x >> 6, D = c.D, C = c.C;
    float dgk[CMAX];
    for (int cc = 0; cc < C; cc++) dgk[cc] = w.dg[((long)n * NF + k) * C + cc];
    double tacc[CMAX];
    for (int i = 0; i < CMAX; i++) tacc[i] = 0;
    for (int px = p0 + wv; px < p1; px += 4) {
        const float f = feat[((long)n * c.P + px) * NF + k];
        float om[DMAX];
        if (D) basis(px, c.w, D, om);
        for (int cc = 0; cc < C; cc++) {
            float v = wave_sum(f * dgk[cc]);
            if (D) {
                const double *mu = w.mu + ((long)n * C + cc) * (2 + D);
                const double *GM = w.GM + ((long)n * C + cc) * 2 * D, *GQ = w.GQ + ((long)n * C + cc) * D * D;
                const double *Gmu = w.Gmu + ((long)n * C + cc) * D;
                const double f0 = (double)w.gt[((long)n * 2) * c.P + px] - mu[0], f1 = (double)w.gt[((long)n * 2 + 1) * c.P + px] - mu[1];
                double s = 0;
                for (int j = 0; j < D; j++) {
                    const double oj = (double)om[j] - mu[2 + j];
                    s += (f0 * GM[j] + f1 * GM[D + j]) * oj + Gmu[j] * (double)om[j];
                    for (int l = 0; l < D; l++) s += oj * GQ[j * D + l] * ((double)om[l] - mu[2 + l]);
                }
                v += (float)s;
            }
            if (k == 0) {
                w.gw[((long)n * C + cc) * c.P + px] = v;
                tacc[cc] += (double)v * (double)w.p[((long)n * C + cc) * c.P + px];
            }
        }
    }
    __shared__ double sh[4][CMAX];
    if (k == 0) for (int cc = 0; cc < C; cc++) sh[wv][cc] = tacc[cc];
    __syncthreads();
    if (threadIdx.x < C) w.part[((long)n * NCHUNK + chunk) * PARTW + threadIdx.x] = sh[0][threadIdx.x] + sh[1][threadIdx.x] + sh[2][threadIdx.x] + sh[3][threadIdx.x];
}
__global__ void bwd_T_final_kernel(Cfg c, Ws w) {
    const int t = threadIdx.x;
    if (t < c.NB * c.C) w.T[t] = w.red[(long)(t / c.C) * PARTW + (t % c.C)] / w.S[t];   // T_c = sum_px (p/S) gw
}

// ------------------------------------------------------------------------------- backward pass 3
// dp -> softmax backward -> dlogits (NHWC, pitch Cp); residual gradient dR (NHWC [B][P][4C], written);
// feature gradient df[px][k] = lrelu'(f) * sum_c dg[k][c] w_c(px).
// A wavefront takes 64 pixels: first every LANE does the per-pixel arithmetic of its own pixel (the first version ran it in
// all 64 lanes of a wavefront per pixel: 0.84 ms per step), then the lanes turn into the 64 features and walk the 64
// pixels, the pixel's C weights w_c broadcast from the lane that owns it.  Same float operations per value.
__global__ void __launch_bounds__(RB) bwd_pixel_kernel(Cfg c, Ws w, const float *__restrict__ feat,
                                                       const float *__restrict__ R, const float *__restrict__ tgt0,
                                                       const float *__restrict__ tgt1, float *__restrict__ dlogits,
                                                       float *__restrict__ dR, float *__restrict__ dfeat) {
    const int n = blockIdx.y;
    const int k = threadIdx.x & 63, wv = threadIdx.x >> 6, D = c.D, C = c.C;
    const int b = n >> 1, d = n & 1;
    float dgk[CMAX];
    for (int cc = 0; cc < C; cc++) dgk[cc] = w.dg[((long)n * NF + k) * C + cc];
    const float ecoef = c.w_entropy / ((float)c.NB * (float)c.P);
    const float tcoef = 1.f / ((float)c.NB * (float)c.P);
    const float ycen = c.w_compact != 0.f ? (float)w.cen[2 * n] : 0.f, xcen = c.w_compact != 0.f ? (float)w.cen[2 * n + 1] : 0.f;
    for (int base = (blockIdx.x * 4 + wv) * 64; base < c.P; base += gridDim.x * 4 * 64) {
        const int px = base + k;                         // phase 1: lane = pixel
        float q[CMAX];                                   // w_c(px) = p_c / S_c
        for (int cc = 0; cc < CMAX; cc++) q[cc] = 0.f;
        if (px < c.P) {
        float pv[CMAX], dp[CMAX];
        const float g0 = w.G[((long)n * 2) * c.P + px], g1 = w.G[((long)n * 2 + 1) * c.P + px];
        float om[DMAX];
        if (D) basis(px, c.w, D, om);
        const float *r = R + ((long)b * c.P + px) * 4 * C + d * 2 * C;
        float *dr = dR + ((long)b * c.P + px) * 4 * C + d * 2 * C;
        float psum = 0.f;
        for (int cc = 0; cc < C; cc++) { pv[cc] = w.p[((long)n * C + cc) * c.P + px]; psum += pv[cc]; }
        float pm = pv[0];
        for (int cc = 1; cc < C; cc++) pm = fmaxf(pm, pv[cc]);
        float se = 0.f;
        for (int cc = 0; cc < C; cc++) se += expf(pv[cc] - pm);
        const float lse = pm + logf(se);
        for (int cc = 0; cc < C; cc++) {
            const float S = (float)w.S[(long)n * C + cc];
            q[cc] = pv[cc] / S;
            float v = g0 * w.u[((long)n * 2) * C + cc] + g1 * w.u[((long)n * 2 + 1) * C + cc];   // agg
            const float r0 = r[cc], r1 = r[C + cc];
            if (c.tanh_res) {
                const float t0 = tanhf(r0 / c.div_coeff), t1 = tanhf(r1 / c.div_coeff);
                v += c.res_scale * (g0 * t0 + g1 * t1);
                dr[cc] = g0 * pv[cc] * c.res_scale * (1.f - t0 * t0) / c.div_coeff;
                dr[C + cc] = g1 * pv[cc] * c.res_scale * (1.f - t1 * t1) / c.div_coeff;
            } else {
                v += g0 * r0 + g1 * r1;
                dr[cc] = g0 * pv[cc];
                dr[C + cc] = g1 * pv[cc];
            }
            if (D) {
                const double *A = w.A + ((long)n * C + cc) * 2 * D;
                const double *mu = w.mu + ((long)n * C + cc) * (2 + D) + 2;
                float s0 = 0.f, s1 = 0.f;
                for (int j = 0; j < D; j++) { const float oj = om[j] - (float)mu[j]; s0 += (float)A[j] * oj; s1 += (float)A[D + j] * oj; }
                v += g0 * s0 + g1 * s1;
            }
            v += (w.gw[((long)n * C + cc) * c.P + px] - (float)w.T[(long)n * C + cc]) / S;           // through w = p / S
            if (c.w_entropy != 0.f) {
                const float s2 = expf(pv[cc] - lse);                                              // softmax(p)_c
                v += -ecoef * (2.f * pv[cc] - lse - s2 * psum);
            }
            dp[cc] = v;
        }
        for (int t = 0; t < c.ntgt; t++) {
            float tv = (t ? tgt1 : tgt0)[(long)n * c.P + px];
            if (c.t_th[t] != -1.f) tv = tv > c.t_th[t] ? 1.f : 0.f;
            const float df = tv - pv[c.tgt_channel];
            dp[c.tgt_channel] += c.t_w[t] * tcoef * (-2.f) * df * (df > 0.f ? c.t_wpos[t] : c.t_wneg[t]);
        }
        if (c.w_compact != 0.f) {
            // d/dm of mean(err * m): the terms through the centroid vanish (sum_q m_q (y_q - y_c) = 0)
            const int y = px / c.w, x = px - y * c.w;
            const float dy = (float)y / (float)c.h - ycen, dx = (float)x / (float)c.w - xcen;
            dp[c.comp_ch] += c.w_compact * tcoef * (dy * dy + dx * dx);
        }
        if (c.sharpen_mode == 1) {
            // target detached; logp = log_softmax(p): d/dp_j = -(t_j - softmax(p)_j * sum t) / (NB*C*P)
            float tt[CMAX], ts = 0.f;
            for (int cc = 0; cc < C; cc++) { tt[cc] = powf(pv[cc], 1.f / c.t_sharpen); ts += tt[cc]; }
            float tsum = 0.f;
            for (int cc = 0; cc < C; cc++) { tt[cc] /= ts; tsum += tt[cc]; }
            for (int cc = 0; cc < C; cc++) dp[cc] -= c.w_sharpen * (tcoef / (float)C) * (tt[cc] - expf(pv[cc] - lse) * tsum);
        } else if (c.sharpen_mode == 2) {
            float mo = 0.f;
            for (int cc = 0; cc < C; cc++) if (cc != c.tgt_channel) mo = fmaxf(mo, pv[cc]);
            const float df = pv[c.tgt_channel] - mo;
            if (c.t_sharpen - fabsf(df) > 0.f) dp[c.tgt_channel] -= c.w_sharpen * tcoef * (df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f));
        }
        {
            float dot = 0.f;
            for (int cc = 0; cc < C; cc++) dot += pv[cc] * dp[cc];
            float *dl = dlogits + ((long)n * c.P + px) * c.Cp;
            for (int cc = 0; cc < C; cc++) dl[cc] = pv[cc] * (dp[cc] - dot);
            for (int cc = C; cc < c.Cp; cc++) dl[cc] = 0.f;
        }
        }
        // phase 2: lane = feature k; pixel base + j's weights come from lane j
        const int npx = min(64, c.P - base);
        for (int j = 0; j < npx; j++) {
            float wsum = 0.f;        // sum_c dg[k][c] w_c(px)
#pragma unroll
            for (int cc = 0; cc < CMAX; cc++)
                if (cc < C) wsum += dgk[cc] * __shfl(q[cc], j, 64);
            const long o = ((long)n * c.P + base + j) * NF + k;
            dfeat[o] = wsum * (feat[o] > 0.f ? 1.f : 0.1f);
        }
    }
}

__global__ void __launch_bounds__(256) lrelu_bwd_kernel(const float *__restrict__ dy, const float *__restrict__ y,
                                                        float *__restrict__ dx, long n4, float slope) {
    const long step = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += step) {
        const f32x4 g = reinterpret_cast<const f32x4 *>(dy)[i], v = reinterpret_cast<const f32x4 *>(y)[i];
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; e++) o[e] = v[e] > 0.f ? g[e] : g[e] * slope;
        reinterpret_cast<f32x4 *>(dx)[i] = o;
    }
}

__global__ void loss_to_float_kernel(const double *__restrict__ in, float *__restrict__ out) {
    if (threadIdx.x < 8) out[threadIdx.x] = (float)in[threadIdx.x];
}
__global__ void scale_kernel(float *__restrict__ g, long n, float sc) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) g[i] *= sc;
}

int make_cfg(const rcf_flowhead_cfg *s, Cfg &c) {
    if (!s || s->B <= 0 || s->C <= 0 || s->C > CMAX || s->h <= 0 || s->w <= 0 || s->nf != NF) return RCF_EINVAL;
    if (s->D != 0 && s->D != 2 && s->D != 5) return RCF_EINVAL;
    if (s->logits_pitch < s->C || s->n_targets < 0 || s->n_targets > 2) return RCF_EINVAL;
    if ((s->n_targets || s->sharpen_mode == 2) && (s->target_channel < 0 || s->target_channel >= s->C)) return RCF_EINVAL;
    if (s->w_compact != 0.f && (s->compact_channel < 0 || s->compact_channel >= s->C)) return RCF_EINVAL;
    if (s->sharpen_mode < 0 || s->sharpen_mode > 2 || (s->sharpen_mode && !(s->t_sharpen > 0.f))) return RCF_EINVAL;
    if (2 * s->B * s->C > 256 || (2 + 2 * s->D) * s->C > PARTW || (2 + s->D) * s->C > PARTW) return RCF_EINVAL;
    c.B = s->B; c.NB = 2 * s->B; c.C = s->C; c.P = s->h * s->w; c.h = s->h; c.w = s->w; c.Cp = s->logits_pitch;
    c.D = s->D; c.robust = s->robust; c.tanh_res = s->tanh_residual;
    c.eps = s->eps; c.q = s->q; c.clamp_t = s->clamp_t; c.res_scale = s->res_scale; c.div_coeff = s->div_coeff;
    c.w_seg = s->w_seg; c.w_entropy = s->w_entropy; c.ntgt = s->n_targets; c.tgt_channel = s->target_channel;
    for (int t = 0; t < 2; t++) { c.t_wpos[t] = s->t_wpos[t]; c.t_wneg[t] = s->t_wneg[t]; c.t_w[t] = s->t_weight[t]; c.t_th[t] = s->t_thresh[t]; }
    c.w_compact = s->w_compact; c.comp_ch = s->compact_channel;
    c.w_sharpen = s->sharpen_mode ? s->w_sharpen : 0.f; c.t_sharpen = s->t_sharpen; c.sharpen_mode = s->sharpen_mode;
    return 0;
}

}  // namespace

extern "C" size_t rcf_flowhead_workspace_bytes(const rcf_flowhead_cfg *s) {
    Cfg c;
    if (make_cfg(s, c)) return 0;
    Ws w;
    return carve(nullptr, c, w);
}

#define FH_SETUP()                                                                            \
    Cfg c;                                                                                    \
    if (int e = make_cfg(s, c)) return e;                                                     \
    Ws w;                                                                                     \
    if (!workspace || workspace_bytes < carve(nullptr, c, w) || !rcf_aligned16(workspace)) return RCF_EWORKSPACE; \
    carve((char *)workspace, c, w);                                                           \
    hipStream_t st = rcf_stream(stream);                                                      \
    const dim3 gch(NCHUNK, c.NB)

extern "C" int rcf_flowhead_prepare_f32(const rcf_flowhead_cfg *s, const float *gt_fw, const float *gt_bw,
                                        float *flow4, void *workspace, size_t workspace_bytes, void *stream) {
    FH_SETUP();
    (void)gch;
    if (!gt_fw || !gt_bw || !flow4) return RCF_EINVAL;
    hipLaunchKernelGGL(prepare_kernel, dim3(rcf_cdiv((long)c.NB * c.P, RB)), dim3(RB), 0, st, c, gt_fw, gt_bw, w.gt, flow4);
    RCF_LAUNCH_CHECK();
    return 0;
}

extern "C" int rcf_flowhead_fwd_f32(const rcf_flowhead_cfg *s, const float *logits, const float *feat,
                                    const float *residual, const float *W1, const float *b1, const float *W2,
                                    const float *b2, const float *target0, const float *target1, float *losses_out,
                                    float *masks_out, float *flow_pred, float *flow_agg, float *flow_adj,
                                    float *flow_aff, void *workspace, size_t workspace_bytes, void *stream) {
    FH_SETUP();
    if (!logits || !feat || !residual || !W1 || !b1 || !W2 || !b2 || !losses_out) return RCF_EINVAL;
    if ((c.ntgt > 0 && !target0) || (c.ntgt > 1 && !target1)) return RCF_EINVAL;
    hipLaunchKernelGGL(softmax_kernel, gch, dim3(RB), 0, st, c, w, logits, target0, target1);
    hipLaunchKernelGGL(reduce_parts_kernel, dim3(c.NB), dim3(128), 0, st, (const double *)w.part, w.red, c.C + 6);
    hipLaunchKernelGGL(softmax_final_kernel, dim3(1), dim3(256), 0, st, c, w);
    hipLaunchKernelGGL(pool_kernel, gch, dim3(RB), 0, st, c, w, feat);
    hipLaunchKernelGGL(pool_final_kernel, dim3(c.NB), dim3(64 * c.C), 0, st, c, w);
    hipLaunchKernelGGL(mlp_kernel, dim3(c.NB), dim3(256), 0, st, c, w, W1, b1, W2, b2);
    if (c.D) {
        hipLaunchKernelGGL(moments1_kernel, gch, dim3(RB), 0, st, c, w);
        hipLaunchKernelGGL(reduce_parts_kernel, dim3(c.NB), dim3(128), 0, st, (const double *)w.part, w.red, c.C * (2 + c.D));
        hipLaunchKernelGGL(moments1_final_kernel, dim3(c.NB), dim3(128), 0, st, c, w);
        for (int cc = 0; cc < c.C; cc++) {
            hipLaunchKernelGGL(moments2_kernel, gch, dim3(RB), 0, st, c, w, cc);
            hipLaunchKernelGGL(reduce_parts_kernel, dim3(c.NB), dim3(128), 0, st, (const double *)w.part, w.red,
                               2 * c.D + c.D * (c.D + 1) / 2);
            hipLaunchKernelGGL(affine_solve_kernel, dim3(1), dim3(64), 0, st, c, w, cc);
        }
    }
    hipLaunchKernelGGL(recon_kernel, gch, dim3(RB), 0, st, c, w, residual, flow_pred, flow_agg, flow_adj, flow_aff);
    hipLaunchKernelGGL(reduce_parts_kernel, dim3(c.NB), dim3(128), 0, st, (const double *)w.part, w.red, 2);
    hipLaunchKernelGGL(recon_final_kernel, dim3(1), dim3(64), 0, st, c, w);
    RCF_LAUNCH_CHECK();
    // losses_out (fp32): seg_fw, seg_bw, entropy, target0, target1  (device -> device conversion kernel-free: tiny copy)
    hipLaunchKernelGGL(loss_to_float_kernel, dim3(1), dim3(64), 0, st, (const double *)w.loss, losses_out);
    if (masks_out) {
        hipError_t e = hipMemcpyAsync(masks_out, w.p, (size_t)c.NB * c.C * c.P * sizeof(float), hipMemcpyDeviceToDevice, st);
        if (e != hipSuccess) return (int)e;
    }
    RCF_LAUNCH_CHECK();
    return 0;
}

extern "C" int rcf_flowhead_bwd_f32(const rcf_flowhead_cfg *s, const float *feat, const float *residual,
                                    const float *W1, const float *W2, const float *target0, const float *target1,
                                    float grad_scale, float *dlogits, float *dresidual, float *dfeat, float *dW1,
                                    float *db1, float *dW2, float *db2, void *workspace, size_t workspace_bytes,
                                    void *stream) {
    FH_SETUP();
    if (!feat || !residual || !W1 || !W2 || !dlogits || !dresidual || !dfeat || !dW1 || !db1 || !dW2 || !db2) return RCF_EINVAL;
    if ((c.ntgt > 0 && !target0) || (c.ntgt > 1 && !target1)) return RCF_EINVAL;
    // an upstream scale (loss.backward(gradient=...)) multiplies every coefficient
    Cfg cs = c;
    cs.w_entropy *= grad_scale;
    cs.w_compact *= grad_scale;
    cs.w_sharpen *= grad_scale;
    for (int t = 0; t < 2; t++) cs.t_w[t] *= grad_scale;
    if (grad_scale != 1.f) {
        const long n = (long)c.NB * 2 * c.P;
        hipLaunchKernelGGL(scale_kernel, dim3(rcf_cdiv(n, 256)), dim3(256), 0, st, w.G, n, grad_scale);
    }
    hipLaunchKernelGGL(bwd_reduce_kernel, gch, dim3(RB), 0, st, cs, w);
    hipLaunchKernelGGL(reduce_parts_kernel, dim3(c.NB), dim3(128), 0, st, (const double *)w.part, w.red, c.C * (2 + 2 * c.D));
    hipLaunchKernelGGL(mlp_bwd_kernel, dim3(c.NB), dim3(256), 0, st, cs, w, W1, W2);
    hipLaunchKernelGGL(pgrad_final_kernel, dim3(rcf_cdiv(NF * NF + 3 * NF + 2, 256)), dim3(256), 0, st, cs, w, dW1, db1, dW2, db2);
    hipLaunchKernelGGL(bwd_gw_kernel, gch, dim3(RB), 0, st, cs, w, feat);
    hipLaunchKernelGGL(reduce_parts_kernel, dim3(c.NB), dim3(128), 0, st, (const double *)w.part, w.red, c.C);
    hipLaunchKernelGGL(bwd_T_final_kernel, dim3(1), dim3(256), 0, st, cs, w);
    hipLaunchKernelGGL(bwd_pixel_kernel, dim3(512, c.NB), dim3(RB), 0, st, cs, w, feat, residual, target0, target1, dlogits,
                       dresidual, dfeat);
    RCF_LAUNCH_CHECK();
    return 0;
}

extern "C" int rcf_lrelu_bwd_f32(const float *dy, const float *y, float *dx, long n, float slope, void *stream) {
    if (!dy || !y || !dx || n <= 0 || n % 4) return RCF_EINVAL;
    long nb = (n / 4 + 255) / 256;
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(lrelu_bwd_kernel, dim3((unsigned)nb), dim3(256), 0, rcf_stream(stream), dy, y, dx, n / 4, slope);
    RCF_LAUNCH_CHECK();
    return 0;
}
