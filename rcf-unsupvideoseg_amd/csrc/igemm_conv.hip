// Convolution forward / data-gradient / weight-gradient as implicit GEMM on the fp32 matrix
// cores of gfx950 (v_mfma_f32_32x32x2_f32: exact fp32, 64 cycles per SIMD, 157 TF/s chip peak).
//
// Stands in for torch.nn.Conv2d (cuDNN) on every conv of the RCF path -- reference call sites:
// models/resnet.py:164-203,565-572, models/res_layer.py:54-60, models/fcn_head.py:100-130,
// models/flow_aggregation_head_with_residual.py:84-93.
//
// Data layout: activations NHWC (pixel pitch may exceed C: channel slices of concat buffers),
// weights [Cout][R][S][Cin].  GEMM view: rows = output pixels, cols = output channels,
// K = (r, s, cin) with cin fastest, so a K-slice of 16 is 64 contiguous bytes of one source pixel.
//
// Workgroup = 256 threads = 4 wave64 in a 2x2 grid; each wave owns MR x NR tiles of 32x32
// (16 accumulator VGPRs each).  Per K-step of 16: global -> registers (float4, zero-filled out of
// bounds: this is where padding / dilation / stride live) -> LDS stored K-major so that the MFMA
// operand read `A[i = lane&31][k = lane>>5]` is one conflict-free ds_read_b32 -> MFMA.  LDS is
// double buffered: the loads of K-step t+1 are in flight while step t is multiplied.
// fp32 MFMA is slow enough (64 cycles for 4 KFLOP) that LDS and VALU addressing hide under it.
//
// Block -> tile mapping is XCD aware: workgroup b runs on XCD b%8 (MI355X_MICROARCH.md), so the
// 8 consecutive ids take 8 different row tiles and ids b, b+8, b+16.. walk the column tiles of the
// same row tile: the activation tile is re-read from that XCD's own L2.
#include <type_traits>
#include "rcf_common.h"

#include <cstdlib>

// No mutable process state in this file: every choice a caller can influence travels in rcf_conv_shape.flags (include/rcf_hip.h
// RCF_CONV_*), per call.

namespace {

constexpr int BK = 16;   // K-step of the weight-gradient kernel

struct IgemmParams {
    const float *A;      // source activations (x for fwd, dy for dgrad)
    const float *Bw;     // weights
    const float *bias;   // per output column or null
    float *Y;
    int M, Ncol, K;
    int Ho, Wo;          // spatial dims of the GEMM-row tensor
    int Hs, Ws;          // spatial dims of the source tensor
    int Cs;              // source channels per tap
    int S;               // kernel width
    int up, off, step, div;  // source coord t = y*up + off + r*step, valid iff t>=0, t%div==0, t/div<Hs
    int a_pitch;
    long a_img_stride;
    int y_pitch;
    int ldb;             // BMODE 0: elements between rows of B[j][k]; BMODE 1: between source channels
    int act;
    float slope;
    int beta;
    int mtiles, ntiles;
    int mtiles8;                  // ceil(mtiles / 8): XCD x (block id % 8) walks the contiguous row tiles [x mtiles8, (x + 1) mtiles8)
    unsigned cs_magic, s_magic;   // floor(2^32/d)+1 for d = Cs, S (exact k/d for k*d < 2^32; 0 when d == 1)
    // split-bf16 / fp16-pair kernels: K order (rcf_common.h rcf_kchunk) -- channel chunk width (Cs = the natural order: one
    // chunk), taps * kch, and their magics
    int kch, rsch;
    unsigned kch_magic, rsch_magic;
    int colmap;                   // rcf_common.h rcf_conv_tile: 1 = an XCD owns column tiles, not a band of row tiles
    int b_bytes;                  // split-bf16 kernels: size of the weight operand (buffer descriptor range)
    unsigned flags;               // rcf_conv_shape.flags of the call (RCF_CONV_*)
    int a_split;                  // conv_h2d_kernel: A holds fp16 pair planes ([pixel][h: C fp16 | m: C fp16], scale from amax_a)
    // split-bf16 kernels: the GEMM rows are the pixels of the rectangle [ry0, ry0+rh) x [rx0, rx0+rw) of every image
    // of the [N, Ho, Wo] row tensor (M = N*rh*rw); the full tensor is the rectangle (0, 0, Ho, Wo)
    int ry0, rx0, rh, rw;
    int rband;                    // > 0: only the frame of this thickness along the rectangle's border
    int rr;                       // rows per image: rh*rw, or the frame's pixel count
    // batched GEMM (rcf_gemm_nt_batched_f32): blockIdx.y = i0 * batch1 + i1 selects the operands of one product
    int batch1;
    long a_bs0, a_bs1, b_bs0, b_bs1, y_bs0, y_bs1;     // element strides of A / B / Y over the two batch indices
    // split-bf16 forward only: per row tile, the fp64 column sums and sums of squares of the values written
    // ([mtiles][2 * Ncol]; the batch-norm statistics of the output without reading it back).  Needs act == beta == 0.
    double *stats;
    // fp16-pair kernels: max |value| of the two operand tensors as raw fp32 bits (device scalars); both operands are
    // scaled by powers of two into fp16's range before they are split, the accumulators are scaled back
    const unsigned *amax_a, *amax_b;
    // fp16-pair kernels: the weight operand already split and scaled by 2^k(amax_b), 16 bytes per 4 consecutive k of
    // a row [Ncol][ldb/4][h0 h1 h2 h3 m0 m1 m2 m3] (the fp32 layout's addresses); null = fp32 weights in Bw, split on
    // the way into LDS
    const void *b_pairs;
    // conv_h2p_kernel (igemm_h2p.inc): the same split weights as plane-separated K-step blocks [K/16][2][Ncol][32 B, halves
    // swizzled like the LDS tile], fetched global -> LDS by DMA; null = that kernel is not eligible
    const void *b_pairs2;
    int h2p_gn;                   // conv_h2p_kernel: workgroups sharing one row range (they split the column tiles)
    unsigned *amax_out;           // optional: max |value written| (raw bits, atomicMax): the range of the next consumer
    // data gradient whose output is the gradient of a batch norm + ReLU's OUTPUT (rcf_conv2d_dgrad_bnsums_f32): that norm's
    // input x ([rows][bn_x_pitch], the rows of Y), its ReLU sign bits ([rows][Ncol / 4], bit e = output 4 j + e was positive) and
    // constants; `stats` then receives per row tile [sum g | sum g xhat], g = the masked value written, xhat = (x - mean) invstd
    const float *bn_x;
    const unsigned char *bn_mask;
    const float *bn_mean, *bn_invstd;
    int bn_x_pitch;
    // data gradient + masked addend (rcf_conv2d_dgrad_add_f32): Y = result + (mask ? add : 0); add [rows][add_pitch], mask
    // [rows][Ncol / 4] -- the identity branch of a residual join, whose gradient is the join's output gradient under its ReLU mask
    const float *add_src;
    const unsigned char *add_mask;
    int add_pitch;
};

// pixel `pix` (0 <= pix < rr) of a region -> image coordinates.  Rectangle: row-major.  Frame of thickness t: the top
// strip (t x rw), the bottom strip, then the left and right strips (each (rh - 2t) x t), all row-major.
__device__ __forceinline__ void region_yx(int pix, int ry0, int rx0, int rh, int rw, int t, int &y, int &x) {
    if (t <= 0) {
        const int yr = pix / rw;
        y = yr + ry0;
        x = pix - yr * rw + rx0;
        return;
    }
    const int strip = t * rw;
    if (pix < 2 * strip) {
        const int bottom = pix >= strip;
        const int q = pix - (bottom ? strip : 0);
        const int yr = q / rw;
        y = ry0 + yr + (bottom ? rh - t : 0);
        x = rx0 + q - yr * rw;
    } else {
        int q = pix - 2 * strip;
        const int side = t * (rh - 2 * t);
        const int right = q >= side;
        q -= right ? side : 0;
        const int yr = q / t;
        y = ry0 + t + yr;
        x = rx0 + q - yr * t + (right ? rw - t : 0);
    }
}

__device__ __forceinline__ int fast_div(int k, unsigned magic) { return magic ? (int)__umulhi((unsigned)k, magic) : k; }

// K-major LDS tiles (X[k][ld]): operand fetch is one conflict-free ds_read_b32 per K=2 MFMA.
template <int MR, int NR, int BKT>
__device__ __forceinline__ void mma_tile(const float *__restrict__ As, const float *__restrict__ Bs, int lda,
                                         int ldb, int wm, int wn, int lane, f32x16 (&acc)[MR][NR]) {
    const int l31 = lane & 31, kh = lane >> 5;
#pragma unroll
    for (int kk = 0; kk < BKT / 2; ++kk) {
        const int krow = 2 * kk + kh;
        float a[MR], b[NR];
#pragma unroll
        for (int mr = 0; mr < MR; ++mr) a[mr] = As[krow * lda + wm * 32 * MR + mr * 32 + l31];
#pragma unroll
        for (int nr = 0; nr < NR; ++nr) b[nr] = Bs[krow * ldb + wn * 32 * NR + nr * 32 + l31];
#pragma unroll
        for (int mr = 0; mr < MR; ++mr)
#pragma unroll
            for (int nr = 0; nr < NR; ++nr)
                acc[mr][nr] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mr], b[nr], acc[mr][nr], 0, 0, 0);
    }
}

// Row-major A tile (A[row][BKT+4]): one ds_read_b128 per lane feeds FOUR K=2 MFMAs -- lanes 0-31 hold
// k = 8g..8g+3, lanes 32-63 k = 8g+4..8g+7 of their row, and MFMA e of the group contracts the pair
// {8g+e, 8g+4+e}.  Any pairing of k is valid as long as B uses the same one.  B is either row-major
// too (B_RM, forward weights) or K-major (dgrad weights: ds_read_b32 at row 8g + 4*kh + e).
template <int MR, int NR, int BKT, bool B_RM>
__device__ __forceinline__ void mma_tile_rm(const float *__restrict__ As, const float *__restrict__ Bs, int ldb,
                                            int wm, int wn, int lane, f32x16 (&acc)[MR][NR]) {
    constexpr int LDK = BKT + 4;
    const int l31 = lane & 31, kh = lane >> 5;
#pragma unroll
    for (int g = 0; g < BKT / 8; ++g) {
        f32x4 av[MR], bv[NR];
#pragma unroll
        for (int mr = 0; mr < MR; ++mr)
            av[mr] = *reinterpret_cast<const f32x4 *>(As + (wm * 32 * MR + mr * 32 + l31) * LDK + g * 8 + kh * 4);
        if (B_RM) {
#pragma unroll
            for (int nr = 0; nr < NR; ++nr)
                bv[nr] = *reinterpret_cast<const f32x4 *>(Bs + (wn * 32 * NR + nr * 32 + l31) * LDK + g * 8 + kh * 4);
        } else {
#pragma unroll
            for (int nr = 0; nr < NR; ++nr)
#pragma unroll
                for (int e = 0; e < 4; ++e) bv[nr][e] = Bs[(g * 8 + kh * 4 + e) * ldb + wn * 32 * NR + nr * 32 + l31];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int mr = 0; mr < MR; ++mr)
#pragma unroll
                for (int nr = 0; nr < NR; ++nr)
                    acc[mr][nr] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mr][e], bv[nr][e], acc[mr][nr], 0, 0, 0);
    }
}

// BMODE 0: B[j][k], k contiguous (forward: w[co][rs*Cin + c]).
// BMODE 1: B[k = rs*Cs + kc][j], j contiguous (dgrad: w[co = kc][rs][c = j]).
// BKT: K-step (16 or 32).  RM: row-major LDS tiles for the k-contiguous operands (A always, B in BMODE 0).
// STRIDED: dgrad of a stride>1 conv (taps must also pass a divisibility test).
template <int MR, int NR, int BMODE, int BKT, bool RM, bool STRIDED>
__global__ void __launch_bounds__(256) igemm_conv_kernel(IgemmParams p) {
    constexpr int BM = 64 * MR, BN = 64 * NR;
    constexpr int LDA = BM + 4, LDB = BN + 4, LDK = BKT + 4;
    constexpr bool B_RM = RM && BMODE == 0;
    constexpr int A_SIZE = RM ? BM * LDK : BKT * LDA;
    constexpr int B_SIZE = B_RM ? BN * LDK : BKT * LDB;
    constexpr int STAGE = A_SIZE + B_SIZE;
    __shared__ __attribute__((aligned(16))) float smem[2 * STAGE];

    const int bid = blockIdx.x;
    const int grp = bid / (8 * p.ntiles);
    const int rem = bid - grp * 8 * p.ntiles;
    const int tile_n = rem >> 3;
    const int tile_m = grp * 8 + (rem & 7);
    if (tile_m >= p.mtiles) return;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // ---- loader geometry for k-contiguous operands: TPR threads cover one row's BKT floats
    constexpr int TPR = BKT / 4, ROWS = 256 / TPR;
    constexpr int A_PASS = BM / ROWS, B_PASS0 = BN / ROWS;
    const int kq = tid % TPR, arow = tid / TPR;
    const float *arowp[A_PASS];      // STRIDED: image base; else pointer of tap (0,0) of the row (may be out of range)
    int ay[A_PASS], ax[A_PASS];
    const int HoWo = p.Ho * p.Wo;
#pragma unroll
    for (int i = 0; i < A_PASS; ++i) {
        const int m = m0 + arow + ROWS * i;
        if (m < p.M) {
            const int n = m / HoWo;
            const int pix = m - n * HoWo;
            const int y = pix / p.Wo;
            const int x = pix - y * p.Wo;
            ay[i] = y * p.up + p.off;
            ax[i] = x * p.up + p.off;
            arowp[i] = p.A + (long)n * p.a_img_stride + (STRIDED ? 0L : ((long)ay[i] * p.Ws + ax[i]) * p.a_pitch);
        } else {
            arowp[i] = p.A;
            ay[i] = -(1 << 28);
            ax[i] = -(1 << 28);
        }
    }
    // ---- BMODE 1 loader geometry (j-contiguous weights)
    constexpr int BTPR = BN / 4, BROWS = 256 / BTPR, B_PASS1 = BKT / BROWS;
    constexpr int B_PASS = BMODE == 0 ? B_PASS0 : B_PASS1;
    const int bjq = tid % BTPR, bkr = tid / BTPR;

    f32x4 ra[A_PASS], rb[B_PASS];
    bool va[A_PASS], vb[B_PASS];        // validity of the loaded quads: the zero-masking happens at LDS-store
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};   // time, so the loads stay in flight under the MFMAs

    // Branch-free loader: an out-of-image / out-of-range tap reads the (always valid) buffer base and is
    // masked to zero, so the whole K-step is ONE basic block and the scheduler can slide the address
    // arithmetic under the MFMAs.  k -> (r, s, c) by multiply-high with precomputed magics.
    auto load_tile = [&](int kt) {
        const int k = kt * BKT + kq * 4;
        const bool kv = k < p.K;
        const int rs = fast_div(k, p.cs_magic);
        const int c = k - rs * p.Cs;
        const int r = fast_div(rs, p.s_magic);
        const int s = rs - r * p.S;
        const int dy = r * p.step, dx = s * p.step;
        const long tapoff = ((long)dy * p.Ws + dx) * p.a_pitch + c;
#pragma unroll
        for (int i = 0; i < A_PASS; ++i) {
            int ty = ay[i] + dy, tx = ax[i] + dx;
            bool v;
            const float *src;
            if (STRIDED) {
                v = kv && ty >= 0 && tx >= 0 && (ty % p.div == 0) && (tx % p.div == 0);
                ty /= p.div;
                tx /= p.div;
                v = v && ty < p.Hs && tx < p.Ws;
                src = arowp[i] + ((long)ty * p.Ws + tx) * p.a_pitch + c;
            } else {
                v = kv && (unsigned)ty < (unsigned)p.Hs && (unsigned)tx < (unsigned)p.Ws;
                src = arowp[i] + tapoff;
            }
            ra[i] = *reinterpret_cast<const f32x4 *>(v ? src : p.A);
            va[i] = v;
        }
        if (BMODE == 0) {
#pragma unroll
            for (int i = 0; i < B_PASS; ++i) {
                const int j = n0 + arow + ROWS * i;
                const bool v = kv && j < p.Ncol;
                rb[i] = *reinterpret_cast<const f32x4 *>(v ? p.Bw + (long)j * p.ldb + k : p.Bw);
                vb[i] = v;
            }
        } else {
#pragma unroll
            for (int i = 0; i < B_PASS; ++i) {
                const int kb = kt * BKT + bkr + i * BROWS;
                const int j = n0 + bjq * 4;
                const bool v = kb < p.K && j < p.Ncol;
                const int rsb = fast_div(kb, p.cs_magic);
                const int kc = kb - rsb * p.Cs;
                rb[i] = *reinterpret_cast<const f32x4 *>(v ? p.Bw + (long)kc * p.ldb + (long)rsb * p.Ncol + j : p.Bw);
                vb[i] = v;
            }
        }
    };
    auto store_tile = [&](int buf) {
        float *As = smem + buf * STAGE;
        float *Bs = As + A_SIZE;
#pragma unroll
        for (int i = 0; i < A_PASS; ++i) ra[i] = va[i] ? ra[i] : zero4;
#pragma unroll
        for (int i = 0; i < B_PASS; ++i) rb[i] = vb[i] ? rb[i] : zero4;
        if (RM) {
#pragma unroll
            for (int i = 0; i < A_PASS; ++i)
                *reinterpret_cast<f32x4 *>(As + (arow + ROWS * i) * LDK + kq * 4) = ra[i];
        } else {
#pragma unroll
            for (int i = 0; i < A_PASS; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) As[(kq * 4 + e) * LDA + arow + ROWS * i] = ra[i][e];
        }
        if (BMODE == 0) {
            if (B_RM) {
#pragma unroll
                for (int i = 0; i < B_PASS; ++i)
                    *reinterpret_cast<f32x4 *>(Bs + (arow + ROWS * i) * LDK + kq * 4) = rb[i];
            } else {
#pragma unroll
                for (int i = 0; i < B_PASS; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e) Bs[(kq * 4 + e) * LDB + arow + ROWS * i] = rb[i][e];
            }
        } else {
#pragma unroll
            for (int i = 0; i < B_PASS; ++i)
                *reinterpret_cast<f32x4 *>(Bs + (bkr + i * BROWS) * LDB + bjq * 4) = rb[i];
        }
    };

    f32x16 acc[MR][NR];
#pragma unroll
    for (int mr = 0; mr < MR; ++mr)
#pragma unroll
        for (int nr = 0; nr < NR; ++nr)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mr][nr][e] = 0.f;

    const int KT = (p.K + BKT - 1) / BKT;
    load_tile(0);
    store_tile(0);
    __syncthreads();
    // steady state is one straight-line block: loads of step t+1, MFMAs of step t, LDS stores of t+1
    for (int kt = 0; kt < KT - 1; ++kt) {
        const int cur = kt & 1;
        load_tile(kt + 1);
        // keep the global loads ABOVE the MFMAs (LLVM otherwise sinks them next to the LDS stores that consume
        // them and exposes the HBM latency on every K-step)
        __builtin_amdgcn_sched_barrier(0);
        const float *As = smem + cur * STAGE;
        if (RM) mma_tile_rm<MR, NR, BKT, B_RM>(As, As + A_SIZE, LDB, wm, wn, lane, acc);
        else mma_tile<MR, NR, BKT>(As, As + A_SIZE, LDA, LDB, wm, wn, lane, acc);
        __builtin_amdgcn_sched_barrier(0);
        store_tile(cur ^ 1);
        __syncthreads();
    }
    {
        const float *As = smem + ((KT - 1) & 1) * STAGE;
        if (RM) mma_tile_rm<MR, NR, BKT, B_RM>(As, As + A_SIZE, LDB, wm, wn, lane, acc);
        else mma_tile<MR, NR, BKT>(As, As + A_SIZE, LDA, LDB, wm, wn, lane, acc);
    }

    // ---- epilogue: lane holds column (lane&31), rows (e&3) + 8*(e>>2) + 4*(lane>>5)
    const int l31 = lane & 31, kh = lane >> 5;
#pragma unroll
    for (int nr = 0; nr < NR; ++nr) {
        const int col = n0 + wn * 32 * NR + nr * 32 + l31;
        if (col >= p.Ncol) continue;
        const float bv = p.bias ? p.bias[col] : 0.f;
#pragma unroll
        for (int mr = 0; mr < MR; ++mr) {
            const int rbase = m0 + wm * 32 * MR + mr * 32 + 4 * kh;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = rbase + (e & 3) + 8 * (e >> 2);
                if (row < p.M) {
                    float *dst = p.Y + (long)row * p.y_pitch + col;
                    float v = acc[mr][nr][e] + bv;
                    if (p.act == 1) v = v > 0.f ? v : v * p.slope;
                    if (p.beta) v += *dst;
                    *dst = v;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------- split-bf16 ("x3")
// fp32 contraction on the bf16 matrix cores.  fp32 MFMA runs at 1/16 of the bf16 rate on gfx950, so
// each fp32 operand is split EXACTLY into three bf16 parts x = h + m + l (round-to-nearest at every
// level: 3 x 8 significand bits cover the 24 of fp32, residuals are exact in fp32) and the product is
// formed from the six partial products whose weight is >= 2^-16:
//     x*y ~= h*h' + (h*m' + m*h') + (m*m' + h*l' + l*h')        (dropped: m*l', l*m', l*l' <= 2^-25 |xy|)
// bf16 x bf16 is exact in fp32 and the accumulation is fp32, so the result carries a relative error per
// product (~2^-25) BELOW fp32's own rounding (2^-24): it is fp32 arithmetic at 6 MFMA passes of 1/16
// cost each = 2.67x the fp32-MFMA rate (peak 2.5 PF / 6 = 417 TF/s).
// The split happens ONCE per element, on the way from global memory into LDS; LDS holds three bf16
// planes per operand, row-major [row][16 k] = 32 B per row; the two 16-byte halves of a row are swapped
// on rows with bit 3 set so that the ds_read_b128 operand fetch (lane = row, lane>>5 = k half) and the
// ds_write_b64 of the loader are both bank-conflict free without padding.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split3(const f32x4 v, u32x2 &h, u32x2 &m, u32x2 &l) {
    const f32x2 a = {v[0], v[1]}, b = {v[2], v[3]};
    const bf16x2 ha = __builtin_convertvector(a, bf16x2), hb = __builtin_convertvector(b, bf16x2);
    const f32x2 ra = a - __builtin_convertvector(ha, f32x2), rb = b - __builtin_convertvector(hb, f32x2);
    const bf16x2 ma = __builtin_convertvector(ra, bf16x2), mb = __builtin_convertvector(rb, bf16x2);
    const f32x2 sa = ra - __builtin_convertvector(ma, f32x2), sb = rb - __builtin_convertvector(mb, f32x2);
    const bf16x2 la = __builtin_convertvector(sa, bf16x2), lb = __builtin_convertvector(sb, bf16x2);
    h = u32x2{__builtin_bit_cast(unsigned, ha), __builtin_bit_cast(unsigned, hb)};
    m = u32x2{__builtin_bit_cast(unsigned, ma), __builtin_bit_cast(unsigned, mb)};
    l = u32x2{__builtin_bit_cast(unsigned, la), __builtin_bit_cast(unsigned, lb)};
}

// -------------------------------------------------------------------------------------------- fp16 pairs ("h2")
// The same idea with HALF the matrix-core work: x * 2^k = h + m with h, m fp16 (2 x 11 significand bits; k brings the
// tensor's largest magnitude to [2^14, 2^15) so that neither part overflows and m stays a normal number for every
// element within 2^-18 of the largest) and x*y ~= h*h' + (h*m' + m*h'): the dropped m*m' and the representation error
// are <= 2^-22 |xy| per product, the level of fp32's own rounding through the accumulation (measured against float64:
// 3.0e-7 relative rms at K = 2304, torch's fp32 GEMM 3.1e-7).  3 MFMA passes: peak 2.5 PF / 3 = 833 TF/s.
// The power-of-two scales are exact; they come from the tensors' max |value| (rcf_absmax_f32, a device scalar that the
// caller can reuse between the forward, data-gradient and weight-gradient launches that share an operand).
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

// 2^k with k = 14 - floor(log2(max)), |k| <= 100 (amax: raw bits of a non-negative fp32)
__device__ __forceinline__ int h2_exponent(unsigned amax_bits) {
    const int e = (int)((amax_bits >> 23) & 0xffu);
    if (amax_bits == 0u) return 0;
    int k = 14 - (e - 127);
    return k > 100 ? 100 : (k < -100 ? -100 : k);
}
__device__ __forceinline__ float pow2f(int k) { return __builtin_bit_cast(float, (unsigned)(127 + k) << 23); }

// x * s = h + m + (rest below 2^-22 |x s|), h and m in fp16: h = fp16(x s) (s is a power of two: the product is exact),
// m = fp16(x s - h) (the difference is exact in fp32).  Two values per register pair through gfx950's mixed-precision FMAs:
// v_fma_mix{lo,hi}_f16 evaluate fma(fp32, fp32, fp32-or-fp16) in fp32 and round the result once to fp16 into the low /
// high half of the destination -- 8 instructions per four values where multiply / convert / convert back / subtract /
// convert took 14 (the weight-gradient kernel, which splits both operands, spent 10 VALU instructions per MFMA).
// `- 0.0` as the addend keeps the sign of a zero product.
__device__ __forceinline__ unsigned split2h_pair(float a, float b, float s, unsigned &m) {
    unsigned h;
    const float nz = -0.0f;
    asm("v_fma_mixlo_f16 %0, %1, %2, %3" : "=v"(h) : "v"(a), "v"(s), "v"(nz));
    asm("v_fma_mixhi_f16 %0, %1, %2, %3" : "+v"(h) : "v"(b), "v"(s), "v"(nz));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(m) : "v"(a), "v"(s), "v"(h));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(m) : "v"(b), "v"(s), "v"(h));
    return h;
}
__device__ __forceinline__ void split2h(const f32x4 v, float s, u32x2 &h, u32x2 &m) {
    unsigned m0, m1;
    const unsigned h0 = split2h_pair(v[0], v[1], s, m0), h1 = split2h_pair(v[2], v[3], s, m1);
    h = u32x2{h0, h1};
    m = u32x2{m0, m1};
}

// amax[0] = max(amax[0], bits(max |x|)) over a [rows][C] matrix with row pitch `pitch`: for non-negative floats the
// raw bits order like the values, so the reduction is an integer max (order independent: deterministic with atomics)
__global__ void __launch_bounds__(256) absmax_kernel(const float *__restrict__ x, long rows, int C, int pitch,
                                                     unsigned *__restrict__ amax) {
    __shared__ unsigned sh[4];
    const int CV = C >> 2;
    const long total = rows * CV, step = (long)gridDim.x * blockDim.x;
    unsigned mx = 0u;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += step) {
        const long r = i / CV;
        const int c4 = (int)(i - r * CV) * 4;
        const f32x4 v = *reinterpret_cast<const f32x4 *>(x + r * pitch + c4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float f = v[e];
            mx = max(mx, __float_as_uint(fabsf(f)));
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = max(mx, (unsigned)__shfl_xor((int)mx, o));
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned m = max(max(sh[0], sh[1]), max(sh[2], sh[3]));       // (only while it raises the running maximum)
        if (m > __hip_atomic_load(amax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(amax, m);
    }
}

// one K=16 step: 6 x MR x NR MFMAs (32x32x16 bf16), smallest partial products first.
// PA / PB: bytes per bf16 plane of the A / B tile; arow0 / brow0: first tile row of this wave.
// PERM: tile rows are stored at x3_prow(row) (the weight-gradient loader's conflict-free write pattern).
__device__ __forceinline__ int x3_prow(int row) { return (row & ~15) | ((row & 3) << 2) | ((row >> 2) & 3); }

// SWAP: the MFMA operands change roles, so the accumulator tile is transposed: a lane then owns ONE tile row of A
// (column lane&31) and, per register quad, FOUR consecutive rows of B -- for the conv kernels one pixel and four
// consecutive output channels, i.e. a 16-byte store (data-gradient epilogue).
template <int MR, int NR, int PA, int PB, bool PERM = false, int NP = 3, bool SWAP = false>
__device__ __forceinline__ void mma_x3(const char *__restrict__ As, const char *__restrict__ Bs, int arow0, int brow0,
                                       int lane, f32x16 (&acc)[MR][NR]) {
    const int l31 = PERM ? x3_prow(lane & 31) : (lane & 31);
    const int swz = ((lane >> 5) ^ ((l31 >> 3) & 1)) << 4;
    if constexpr (NP == 2) {                 // fp16 pairs: m*h', h*m', h*h'
        f16x8 a[MR][2], b[NR][2];
#pragma unroll
        for (int mr = 0; mr < MR; ++mr)
#pragma unroll
            for (int q = 0; q < 2; ++q)
                a[mr][q] = *reinterpret_cast<const f16x8 *>(As + q * PA + (arow0 + mr * 32 + l31) * 32 + swz);
#pragma unroll
        for (int nr = 0; nr < NR; ++nr)
#pragma unroll
            for (int q = 0; q < 2; ++q)
                b[nr][q] = *reinterpret_cast<const f16x8 *>(Bs + q * PB + (brow0 + nr * 32 + l31) * 32 + swz);
        constexpr int HA[3] = {1, 0, 0}, HB[3] = {0, 1, 0};
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int mr = 0; mr < MR; ++mr)
#pragma unroll
                for (int nr = 0; nr < NR; ++nr)
                    acc[mr][nr] = SWAP ? __builtin_amdgcn_mfma_f32_32x32x16_f16(b[nr][HB[t]], a[mr][HA[t]], acc[mr][nr], 0, 0, 0)
                                       : __builtin_amdgcn_mfma_f32_32x32x16_f16(a[mr][HA[t]], b[nr][HB[t]], acc[mr][nr], 0, 0, 0);
        return;
    }
    bf16x8 a[MR][3], b[NR][3];
#pragma unroll
    for (int mr = 0; mr < MR; ++mr)
#pragma unroll
        for (int q = 0; q < 3; ++q)
            a[mr][q] = *reinterpret_cast<const bf16x8 *>(As + q * PA + (arow0 + mr * 32 + l31) * 32 + swz);
#pragma unroll
    for (int nr = 0; nr < NR; ++nr)
#pragma unroll
        for (int q = 0; q < 3; ++q)
            b[nr][q] = *reinterpret_cast<const bf16x8 *>(Bs + q * PB + (brow0 + nr * 32 + l31) * 32 + swz);
    constexpr int QA[6] = {2, 0, 1, 1, 0, 0}, QB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int mr = 0; mr < MR; ++mr)
#pragma unroll
            for (int nr = 0; nr < NR; ++nr)
                acc[mr][nr] = SWAP ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[nr][QB[t]], a[mr][QA[t]], acc[mr][nr], 0, 0, 0)
                                   : __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mr][QA[t]], b[nr][QB[t]], acc[mr][nr], 0, 0, 0);
}

constexpr unsigned X3_OOB = 0x80000000u;      // byte offset beyond every descriptor's num_records: loads return 0
// `off` if ok == 1, an out-of-range offset if ok == 0: arithmetic (a select on a load's address tends to become a branch)
__device__ __forceinline__ unsigned x3_oob_unless(unsigned off, int ok) { return (off & ~X3_OOB) | ((unsigned)(ok - 1) & X3_OOB); }

// The epilogue of the transposed-accumulator conv kernels (igemm_conv_x3_kernel, conv_h2d_kernel): a lane owns pixel lane&31 of
// each row tile and, per register quad, output channels 8g + 4(lane>>5) .. +3 of each column tile -- 16-byte stores, the region walk
// once per pixel; scaling back by the operands' powers of two (fp16 pairs), bias / activation / accumulate, the fused batch-norm
// statistics (column sums over pixels = lanes: a reduce-scatter butterfly) and the output's range.  `smem`: the kernel's LDS,
// free once every wave is past its K-loop (the statistics fold reuses it).
template <int MR, int NR, int WM, int WN, bool DGRAD, int NP, bool BST = false>
__device__ __forceinline__ void conv_epilogue_tr(const IgemmParams &p, f32x16 (&acc)[MR][NR], char *smem, int tile_m, int m0, int n0,
                                                 int ka, int kb) {
    constexpr int NT = 64 * WM * WN, BN = 32 * NR * WN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave - wm * WN;
    const int arow0 = wm * 32 * MR, brow0 = wn * 32 * NR;
    const int HoWo = p.rr;
    const int l31 = lane & 31, kh = lane >> 5;
    const bool full = p.rh == p.Ho && p.rw == p.Wo && p.rband <= 0;   // rows map linearly onto the output tensor
    const bool want_stats = !DGRAD && p.stats != nullptr;
    // BST: the batch-norm backward's two sums (lean path only: the launch checks) -- its own instantiation: the sums' registers
    // (32 accumulators beside the tile's 128) would cost every other data gradient occupancy or spills
    constexpr bool want_bstats = DGRAD && BST;
    const float inv_a = pow2f(-ka), inv_b = pow2f(-kb);
    unsigned tmax = 0u;
    // transposed accumulators (mma_x3 SWAP): lane = pixel (lane&31) of each row tile, registers 4g..4g+3 = output
    // channels 8g + 4(lane>>5) .. +3 of each column tile: 16-byte stores, the region walk once per pixel
    long lin[MR];
    bool rowok[MR];
#pragma unroll
    for (int mr = 0; mr < MR; ++mr) {
        const int row = m0 + arow0 + mr * 32 + l31;
        rowok[mr] = row < p.M;
        lin[mr] = row;
        if (rowok[mr] && !full) {
            const int n = row / HoWo;
            int y, x;
            region_yx(row - n * HoWo, p.ry0, p.rx0, p.rh, p.rw, p.rband, y, x);
            lin[mr] = ((long)n * p.Ho + y) * p.Wo + x;
        }
    }
    double *red = reinterpret_cast<double *>(smem);       // [WM][BN][2] (fused batch-norm statistics)
    if (want_stats || want_bstats) __syncthreads();       // every wave is done with the operand stages
    // The common case -- a whole column tile, no bias, no activation (every conv -> batch norm pair and every data
    // gradient of the step) -- as straight-line code: the general loop below tests columns, bias, activation and beta
    // per quad and per element, which the compiler turns into ~170 instructions in four basic blocks per 16-byte
    // store; with 32 quads per thread that was three times the instructions of a 16-step K-loop (the 1x1 layers).
    const bool lean = NP == 2 && p.bias == nullptr && p.act == 0 && n0 + BN <= p.Ncol;
    const bool nts = (p.flags & RCF_CONV_NT_STORES) != 0;      // RCF_CONV_NT_STORES: the tile's output does not stay in L2
#pragma unroll
    for (int nr = 0; nr < NR; ++nr) {
        float cs[16], cq[16];                             // this lane's pixels: sums / sums of squares per channel register
#pragma unroll
        for (int e = 0; e < 16; ++e) { cs[e] = 0.f; cq[e] = 0.f; }
        if (lean) {
            // BETA: accumulate into the output; EXTRA: the by-products are wanted (batch-norm statistics of a forward
            // conv, the output's range of a ViT GEMM) -- a data gradient wants neither: scale, (add,) store
            // EXTRA 2 (data gradients): the value written is the gradient of a batch norm + ReLU's output -- mask it with the
            // norm's sign bits and add g and g * xhat per channel (what rcf_bn_bwd_reduce_mp would read the tensor back for)
            auto quads = [&](auto BETA_, auto EXTRA_) {
                constexpr int BETA = decltype(BETA_)::value;     // 0 overwrite, 1 accumulate into Y, 2 add the masked tensor p.add_src
                constexpr int EXTRA = decltype(EXTRA_)::value;
                const int cb = n0 + brow0 + 4 * kh + nr * 32;           // this lane's first channel of the column tile
#pragma unroll
                for (int mr = 0; mr < MR; ++mr) {
                    if (!rowok[mr]) continue;
                    f32x4 *dst = reinterpret_cast<f32x4 *>(p.Y + lin[mr] * p.y_pitch + cb);
                    const f32x4 *as = nullptr;
                    const unsigned char *am = nullptr;
                    if constexpr (BETA == 2) {
                        as = reinterpret_cast<const f32x4 *>(p.add_src + lin[mr] * p.add_pitch + cb);
                        am = p.add_mask + lin[mr] * (p.Ncol >> 2) + (cb >> 2);
                    }
                    const f32x4 *xs = nullptr;
                    const unsigned char *ms = nullptr;
                    if constexpr (EXTRA == 2) {
                        xs = reinterpret_cast<const f32x4 *>(p.bn_x + lin[mr] * p.bn_x_pitch + cb);
                        ms = p.bn_mask + lin[mr] * (p.Ncol >> 2) + (cb >> 2);
                    }
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        f32x4 v = {acc[mr][nr][4 * g], acc[mr][nr][4 * g + 1], acc[mr][nr][4 * g + 2], acc[mr][nr][4 * g + 3]};
                        v = (v * inv_a) * inv_b;
                        if constexpr (BETA == 1) v += dst[2 * g];
                        if constexpr (BETA == 2) {
                            const f32x4 a = as[2 * g];
                            const unsigned m = am[2 * g];
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] += (m >> e) & 1u ? a[e] : 0.f;   // = the value rcf_bn_bwd_apply_mp's dres would hold
                        }
                        if (nts) __builtin_nontemporal_store(v, dst + 2 * g);
                        else dst[2 * g] = v;
                        if constexpr (EXTRA == 1) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                tmax = max(tmax, __float_as_uint(fabsf(v[e])));
                                if constexpr (!DGRAD) {                  // a data gradient wants the range only
                                    cs[4 * g + e] += v[e];
                                    cq[4 * g + e] = fmaf(v[e], v[e], cq[4 * g + e]);
                                }
                            }
                        }
                        if constexpr (EXTRA == 2) {
                            // the channel constants are re-read per quad (L1 / scalar-cache hits): kept in registers over the
                            // tile they would be 32 more live values
                            const f32x4 mu = *reinterpret_cast<const f32x4 *>(p.bn_mean + cb + 8 * g);
                            const f32x4 is = *reinterpret_cast<const f32x4 *>(p.bn_invstd + cb + 8 * g);
                            const f32x4 xq = xs[2 * g];
                            const unsigned mq = ms[2 * g];
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                tmax = max(tmax, __float_as_uint(fabsf(v[e])));
                                const float gg = (mq >> e) & 1u ? v[e] : 0.f;
                                cs[4 * g + e] += gg;
                                cq[4 * g + e] += gg * ((xq[e] - mu[e]) * is[e]);      // BwdOp's operation order (csrc/bn.hip)
                            }
                        }
                    }
                }
            };
            const bool extra = want_stats || p.amax_out != nullptr;
            using I0 = std::integral_constant<int, 0>;
            using I1 = std::integral_constant<int, 1>;
            using I2 = std::integral_constant<int, 2>;
            const int bmode = (DGRAD && p.add_src) ? 2 : (p.beta ? 1 : 0);
            if constexpr (want_bstats) {
                if (bmode == 2) quads(I2{}, I2{}); else if (bmode == 1) quads(I1{}, I2{}); else quads(I0{}, I2{});
            } else if constexpr (DGRAD) {
                if (bmode == 2) { if (extra) quads(I2{}, I1{}); else quads(I2{}, I0{}); }
                else if (bmode == 1) { if (extra) quads(I1{}, I1{}); else quads(I1{}, I0{}); }
                else { if (extra) quads(I0{}, I1{}); else quads(I0{}, I0{}); }
            } else {
                if (p.beta) { if (extra) quads(I1{}, I1{}); else quads(I1{}, I0{}); }
                else { if (extra) quads(I0{}, I1{}); else quads(I0{}, I0{}); }
            }
        } else {
#pragma unroll
        for (int mr = 0; mr < MR; ++mr) {
            if (!rowok[mr]) continue;
            float *drow = p.Y + lin[mr] * p.y_pitch + n0 + brow0 + 4 * kh;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = n0 + brow0 + nr * 32 + 8 * g + 4 * kh;
                if (c >= p.Ncol) continue;
                f32x4 v = {acc[mr][nr][4 * g], acc[mr][nr][4 * g + 1], acc[mr][nr][4 * g + 2], acc[mr][nr][4 * g + 3]};
                if constexpr (NP == 2) v = (v * inv_a) * inv_b;
                const bool whole = c + 3 < p.Ncol;               // conv channels come in quads; a GEMM's N need not
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (p.bias && (whole || c + e < p.Ncol)) v[e] += p.bias[c + e];
                    if (p.act == 1) v[e] = v[e] > 0.f ? v[e] : v[e] * p.slope;
                    else if (p.act == 2) v[e] = 0.5f * v[e] * (1.f + erff(v[e] * 0.70710678118654752440f));
                }
                float *dq = drow + nr * 32 + 8 * g;
                if (whole) {
                    f32x4 *dst = reinterpret_cast<f32x4 *>(dq);
                    if (p.beta) v += *dst;
                    *dst = v;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (c + e < p.Ncol) {
                            if (p.beta) v[e] += dq[e];
                            dq[e] = v[e];
                        } else {
                            v[e] = 0.f;
                        }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    tmax = max(tmax, __float_as_uint(fabsf(v[e])));
                    cs[4 * g + e] += v[e];
                    cq[4 * g + e] = fmaf(v[e], v[e], cq[4 * g + e]);
                }
            }
        }
        }
        if (want_stats || want_bstats) {                  // block-uniform
            // column sums over the 32 pixels (lanes) of this half-wavefront: a reduce-scatter butterfly -- at every
            // step a lane keeps the half of its registers its lane bit selects and adds the partner's copy of them
            // (16 + 8 + 4 + 2 + 1 values move instead of 5 x 16); bit 0 of the lane ends up redundant.
            const bool b4 = lane & 16, b3 = lane & 8, b2 = lane & 4, b1 = lane & 2;
            float s8[8], q8[8], s4[4], q4[4], s2[2], q2[2];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                s8[i] = (b4 ? cs[8 + i] : cs[i]) + __shfl_xor(b4 ? cs[i] : cs[8 + i], 16);
                q8[i] = (b4 ? cq[8 + i] : cq[i]) + __shfl_xor(b4 ? cq[i] : cq[8 + i], 16);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                s4[i] = (b3 ? s8[4 + i] : s8[i]) + __shfl_xor(b3 ? s8[i] : s8[4 + i], 8);
                q4[i] = (b3 ? q8[4 + i] : q8[i]) + __shfl_xor(b3 ? q8[i] : q8[4 + i], 8);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                s2[i] = (b2 ? s4[2 + i] : s4[i]) + __shfl_xor(b2 ? s4[i] : s4[2 + i], 4);
                q2[i] = (b2 ? q4[2 + i] : q4[i]) + __shfl_xor(b2 ? q4[i] : q4[2 + i], 4);
            }
            float s1 = (b1 ? s2[1] : s2[0]) + __shfl_xor(b1 ? s2[0] : s2[1], 2);
            float q1 = (b1 ? q2[1] : q2[0]) + __shfl_xor(b1 ? q2[0] : q2[1], 2);
            s1 += __shfl_xor(s1, 1);
            q1 += __shfl_xor(q1, 1);
            if ((lane & 1) == 0) {
                const int r = (b4 ? 8 : 0) + (b3 ? 4 : 0) + (b2 ? 2 : 0) + (b1 ? 1 : 0);     // accumulator register = channel
                const int ch = brow0 + nr * 32 + 8 * (r >> 2) + 4 * kh + (r & 3);
                red[(wm * BN + ch) * 2] = (double)s1;
                red[(wm * BN + ch) * 2 + 1] = (double)q1;
            }
        }
    }
    if (p.amax_out) {                                     // block-uniform
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) tmax = max(tmax, (unsigned)__shfl_xor((int)tmax, o));
        if (lane == 0 && tmax > __hip_atomic_load(p.amax_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(p.amax_out, tmax);
    }
    if (want_stats || want_bstats) {
        __syncthreads();
        for (int c = tid; c < BN; c += NT) {
            if (n0 + c >= p.Ncol) continue;
            double sv = 0, qv = 0;
#pragma unroll
            for (int w = 0; w < WM; ++w) {
                sv += red[(w * BN + c) * 2];
                qv += red[(w * BN + c) * 2 + 1];
            }
            double *o = p.stats + (long)tile_m * 2 * p.Ncol + n0 + c;
            o[0] = sv;
            o[p.Ncol] = qv;
        }
    }
}

// forward / dgrad with k-contiguous weights B[j][k] (dgrad: the [Cin][R][S][Cout] transposed copy).
// Workgroup = WM x WN waves, each owning MR x NR accumulator tiles of 32x32: tile (32 MR WM) x (32 NR WN).
// Global loads go through buffer descriptors (32-bit byte offsets; an invalid tap / row / K-tail gets an
// out-of-range offset and the hardware returns zeros: no select on the data, no 64-bit address math).
// DGRAD only names the instantiation (forward and data-gradient launches show up as different kernels in a profile:
// the data gradients run beside the weight gradients of a second stream, the forward convs run alone)
// NP: 3 = bf16 triples (6 partial products), 2 = fp16 pairs (3 partial products, operands scaled by p.amax_a / p.amax_b)
// PRE (NP == 2 only): the weight operand arrives split (p.b_pairs: two fp16 planes)
// TR: transposed accumulator tiles (mma_x3 SWAP) and the 16-byte epilogue; every instance is launched with TR = true
// (the batch-norm statistics, column sums, are a butterfly over the pixels = lanes of a half-wavefront there)
template <int MR, int NR, int WM, int WN, bool STRIDED, bool DGRAD = false, int NP = 3, bool PRE = false, bool TR = DGRAD, bool BST = false>
__global__ void __launch_bounds__(64 * WM * WN, (MR * NR >= 8 && WM * WN == 4) ? 2 : 1) igemm_conv_x3_kernel(IgemmParams p) {
    constexpr int NT = 64 * WM * WN;
    constexpr int BM = 32 * MR * WM, BN = 32 * NR * WN, BKT = 16;
    constexpr int PA = BM * 32, PB = BN * 32, STAGE = NP * (PA + PB);
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];

    // XCD aware: block id % 8 is the XCD; its blocks b, b + 8, .. walk the column tiles of one row tile, then the next row tile
    // of the XCD's own contiguous range (the rows a dilated tap reaches belong to neighbouring row tiles: same L2;
    // HBM traffic -2 ... -5 % in fp32, -14 ... -18 % in bf16 against row tiles interleaved over the XCDs, same time) -- or,
    // p.colmap, the XCD's own column tiles of every row tile (rcf_common.h rcf_conv_tile)
    int tile_m, tile_n;
    rcf_conv_tile((int)blockIdx.x, p.mtiles8, p.ntiles, p.colmap, tile_m, tile_n);
    if (tile_m >= p.mtiles) return;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    if (p.batch1 > 0) {                                       // one product of a batch per grid row
        const int i0 = blockIdx.y / p.batch1, i1 = blockIdx.y - i0 * p.batch1;
        p.A += i0 * p.a_bs0 + i1 * p.a_bs1;
        p.Bw += i0 * p.b_bs0 + i1 * p.b_bs1;
        p.Y += i0 * p.y_bs0 + i1 * p.y_bs1;
    }

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave - wm * WN;

    constexpr int TPR = BKT / 4, ROWS = NT / TPR;            // 4 threads per 16-float row
    constexpr int A_PASS = BM / ROWS, B_PASS = BN / ROWS;
    static_assert(A_PASS >= 1 && B_PASS >= 1, "tile smaller than one loader pass");
    const int kq = tid % TPR, arow = tid / TPR;
    const int st_off = arow * 32 + ((((kq >> 1) ^ ((arow >> 3) & 1))) << 4) + (kq & 1) * 8;   // + ROWS*32 per pass

    const int HoWo = p.rr;                                    // rows per image (the region's pixels)
    const int n_first = m0 / HoWo;                            // first image this tile touches (block-uniform)
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(p.A + (long)n_first * p.a_img_stride), 0, (int)X3_OOB, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(
        (NP == 2 && PRE) ? const_cast<void *>(p.b_pairs) : (void *)const_cast<float *>(p.Bw), 0, p.b_bytes, 0x00020000);

    int abase[A_PASS];        // float offset of tap (0,0) of the row from the descriptor base (STRIDED: of the image)
    int ay[A_PASS], ax[A_PASS];
    // 1x1, stride 1, whole tensor (the bottlenecks' conv1 / conv3 and their data gradients: short K-loops, where every
    // instruction of the tile's set-up counts): GEMM row m IS source pixel m -- no (image, y, x) decomposition, no divisions
    const bool pointwise = !STRIDED && p.S == 1 && p.K == p.Cs && p.up == 1 && p.off == 0 && p.rband <= 0 && p.rh == p.Ho &&
                           p.rw == p.Wo && p.Ho == p.Hs && p.Wo == p.Ws && p.a_img_stride == (long)HoWo * p.a_pitch;
#pragma unroll
    for (int i = 0; i < A_PASS; ++i) {
        const int m = m0 + arow + ROWS * i;
        if (pointwise) {
            const bool ok = m < p.M;
            abase[i] = ok ? (m - n_first * HoWo) * p.a_pitch : 0;
            ay[i] = ok ? 0 : -(1 << 28);
            ax[i] = ok ? 0 : -(1 << 28);
        } else if (m < p.M) {
            const int n = m / HoWo;
            int y, x;
            region_yx(m - n * HoWo, p.ry0, p.rx0, p.rh, p.rw, p.rband, y, x);
            ay[i] = y * p.up + p.off;
            ax[i] = x * p.up + p.off;
            abase[i] = (n - n_first) * (int)p.a_img_stride + (STRIDED ? 0 : (ay[i] * p.Ws + ax[i]) * p.a_pitch);
        } else {
            abase[i] = 0;
            ay[i] = -(1 << 28);
            ax[i] = -(1 << 28);
        }
    }
    // fp16 pairs with the weights split beforehand: same addresses, 16 bytes = (h0..h3, m0..m3) of four k
    constexpr bool pre = NP == 2 && PRE;
    unsigned bbase[B_PASS];   // byte offset of the weight row (out of range for columns past Ncol)
#pragma unroll
    for (int i = 0; i < B_PASS; ++i) {
        const int j = n0 + arow + ROWS * i;
        bbase[i] = j < p.Ncol ? (unsigned)j * (pre ? 64u : (unsigned)p.ldb * 4u) : X3_OOB;
    }
    // A (activations) comes from HBM: its loads run TWO K-steps ahead (two register sets, ping-pong by the parity
    // of the step); B (weights, L2 resident) one step ahead.
    f32x4 ra[2][A_PASS], rb[B_PASS];

    auto load_a = [&](int kt, f32x4 (&dst)[A_PASS]) {
        const int k = kt * BKT + kq * 4;
        const bool kv = k < p.K;
        // position k of the K loop = (channel chunk q, tap rs, channel inside the chunk); natural order: ONE chunk of Cs
        const int q = fast_div(k, p.rsch_magic);
        const int rem = k - q * p.rsch;
        const int rs = fast_div(rem, p.kch_magic);
        const int c = q * p.kch + (rem - rs * p.kch);
        const int r = fast_div(rs, p.s_magic);
        const int s = rs - r * p.S;
        const int dy = r * p.step, dx = s * p.step;
        const int tapoff = (dy * p.Ws + dx) * p.a_pitch + c;
#pragma unroll
        for (int i = 0; i < A_PASS; ++i) {
            int ty = ay[i] + dy, tx = ax[i] + dx;
            int v, off;                      // v: 0 / 1, combined with bitwise ops (no short-circuit control flow)
            if (STRIDED) {
                v = (int)kv & (int)(ty >= 0) & (int)(tx >= 0) & (int)(ty % p.div == 0) & (int)(tx % p.div == 0);
                ty /= p.div;
                tx /= p.div;
                v &= (int)(ty < p.Hs) & (int)(tx < p.Ws);
                off = abase[i] + (ty * p.Ws + tx) * p.a_pitch + c;
            } else {
                v = (int)kv & (int)((unsigned)ty < (unsigned)p.Hs) & (int)((unsigned)tx < (unsigned)p.Ws);
                off = abase[i] + tapoff;
            }
            // invalid -> offset with bit 31 set (beyond num_records): the load returns zeros without touching memory
            const unsigned bo = (((unsigned)off * 4u) & ~X3_OOB) | ((unsigned)(v - 1) & X3_OOB);
            dst[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, (int)bo, 0, 0));
        }
    };
    auto load_b = [&](int kt) {
        const int k = kt * BKT + kq * 4;
        // pre-split weights: K-step major in the loop's own K order, 64 bytes per row and step (pairs_index); raw fp32 weights:
        // the loop position's natural index tap * Cs + c
        unsigned k4;
        if constexpr (pre) {
            k4 = (unsigned)kt * (unsigned)p.Ncol * 64u + (unsigned)kq * 16u;
        } else {
            const int q = fast_div(k, p.rsch_magic);
            const int rem = k - q * p.rsch;
            const int rs = fast_div(rem, p.kch_magic);
            k4 = (unsigned)(rs * p.Cs + q * p.kch + (rem - rs * p.kch)) * 4u;
        }
        const unsigned koob = (unsigned)((int)(k < p.K) - 1) & X3_OOB;
#pragma unroll
        for (int i = 0; i < B_PASS; ++i)
            rb[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB, (int)((bbase[i] + k4) | koob), 0, 0));
    };
    int ka = 0, kb = 0;                      // fp16 pairs: operand scales 2^ka, 2^kb
    if constexpr (NP == 2) {
        ka = h2_exponent(*p.amax_a);
        kb = h2_exponent(*p.amax_b);
    }
    const float sa = pow2f(ka), sb = pow2f(kb);
    auto store_tile = [&](int buf, const f32x4 (&src)[A_PASS]) {
        char *As = smem + buf * STAGE;
        char *Bs = As + NP * PA;
#pragma unroll
        for (int i = 0; i < A_PASS; ++i) {
            u32x2 h, m, l;
            char *d = As + st_off + i * ROWS * 32;
            if constexpr (NP == 2) {
                split2h(src[i], sa, h, m);
            } else {
                split3(src[i], h, m, l);
                *reinterpret_cast<u32x2 *>(d + 2 * PA) = l;
            }
            *reinterpret_cast<u32x2 *>(d) = h;
            *reinterpret_cast<u32x2 *>(d + PA) = m;
        }
#pragma unroll
        for (int i = 0; i < B_PASS; ++i) {
            u32x2 h, m, l;
            char *d = Bs + st_off + i * ROWS * 32;
            if constexpr (pre) {
                const u32x4 q = __builtin_bit_cast(u32x4, rb[i]);
                h = u32x2{q[0], q[1]};
                m = u32x2{q[2], q[3]};
            } else if constexpr (NP == 2) {
                split2h(rb[i], sb, h, m);
            } else {
                split3(rb[i], h, m, l);
                *reinterpret_cast<u32x2 *>(d + 2 * PB) = l;
            }
            *reinterpret_cast<u32x2 *>(d) = h;
            *reinterpret_cast<u32x2 *>(d + PB) = m;
        }
    };

    f32x16 acc[MR][NR];
#pragma unroll
    for (int mr = 0; mr < MR; ++mr)
#pragma unroll
        for (int nr = 0; nr < NR; ++nr)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mr][nr][e] = 0.f;

    const int arow0 = wm * 32 * MR, brow0 = wn * 32 * NR;
    const int KT = (p.K + BKT - 1) / BKT;
    load_a(0, ra[0]);
    load_b(0);
    load_a(1, ra[1]);                        // past the end of K: out-of-range offsets, zeros
    store_tile(0, ra[0]);
    __syncthreads();
    // K-step kt: MFMAs on stage kt&1 while B(kt+1) and A(kt+2) are in flight; then A(kt+1) (loaded a step ago) and
    // B(kt+1) are split into stage (kt+1)&1.  B is issued before A so that its wait does not cover the new A loads.
    int kt = 0;
    for (; kt + 2 <= KT - 1; kt += 2) {
        {   // even step: next A tile is ra[1], the set freed by this step's LDS store is ra[0]
            load_b(kt + 1);
            load_a(kt + 2, ra[0]);
            __builtin_amdgcn_sched_barrier(0);
            const char *As = smem;
            mma_x3<MR, NR, PA, PB, false, NP, TR>(As, As + NP * PA, arow0, brow0, lane, acc);
            store_tile(1, ra[1]);
            __syncthreads();
        }
        {   // odd step
            load_b(kt + 2);
            load_a(kt + 3, ra[1]);
            __builtin_amdgcn_sched_barrier(0);
            const char *As = smem + STAGE;
            mma_x3<MR, NR, PA, PB, false, NP, TR>(As, As + NP * PA, arow0, brow0, lane, acc);
            store_tile(0, ra[0]);
            __syncthreads();
        }
    }
    if (kt < KT - 1) {                       // one more full step (stage 0 -> stage 1)
        load_b(kt + 1);
        __builtin_amdgcn_sched_barrier(0);
        const char *As = smem;
        mma_x3<MR, NR, PA, PB, false, NP, TR>(As, As + NP * PA, arow0, brow0, lane, acc);
        store_tile(1, ra[1]);
        __syncthreads();
        ++kt;
    }
    {
        const char *As = smem + (kt & 1) * STAGE;
        mma_x3<MR, NR, PA, PB, false, NP, TR>(As, As + NP * PA, arow0, brow0, lane, acc);
    }

    conv_epilogue_tr<MR, NR, WM, WN, DGRAD, NP, BST>(p, acc, smem, tile_m, m0, n0, ka, kb);
    static_assert(TR, "the column-per-lane epilogue was retired: every instance runs transposed");
}

// wt[c][rs][co] = w[co][rs][c]: the k-contiguous weight operand of the data gradient
__global__ void __launch_bounds__(256) weight_transpose_kernel(const float *__restrict__ w, float *__restrict__ wt,
                                                               int Cout, int Cin, int RS) {
    __shared__ float tile[32][33];
    const int rs = blockIdx.z, c0 = blockIdx.x * 32, o0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const int co = o0 + r, c = c0 + tx;
        tile[r][tx] = (co < Cout && c < Cin) ? w[((long)co * RS + rs) * Cin + c] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r, co = o0 + tx;
        if (c < Cin && co < Cout) wt[((long)c * RS + rs) * Cout + co] = tile[tx][r];
    }
}

// The weight operand of the fp16-pair kernels, split once per launch instead of once per row tile, in the order the
// kernel reads it: K-step major, [k/16][row][quad of 4 k][h0 h1 h2 h3 m0 m1 m2 m3] (fp16 h, m of w * 2^k, k from the
// weights' range).  The 64 bytes a row contributes to one K-step sit next to the neighbouring rows' 64 bytes, so a
// K-step's weight tile is one contiguous run of full cache lines (row-major fp32 weights give half-used lines whose
// other half is needed a K-step later, after the L1 has been flushed by the activations).
// Element (row j, k) lives at half-index pairs_index(j, k, rows).  TRANSPOSE: rows = c, k = rs * Cout + co.
__device__ __forceinline__ long pairs_index(int j, int k, int rows) {
    return ((((long)(k >> 4) * rows + j) * 4 + ((k >> 2) & 3)) * 8) + (k & 3);
}

template <bool TRANSPOSE>
__global__ void __launch_bounds__(256) weight_pairs_kernel(const float *__restrict__ w, const unsigned *__restrict__ amax,
                                                           _Float16 *__restrict__ planes, int Cout, int Cin, int RS, int korder) {
    const float sc = pow2f(h2_exponent(*amax));
    const long n = (long)Cout * RS * Cin;
    const int Cs = TRANSPOSE ? Cout : Cin, kch = rcf_kchunk(korder, RS, Cs, RCF_KCHUNK_F32);      // K order of the kernel that reads `planes`
    if (!TRANSPOSE) {
        const long step = (long)gridDim.x * blockDim.x;
        const int K = RS * Cin;
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step) {
            const float v = w[i] * sc;
            const _Float16 h = (_Float16)v;
            const int j = (int)(i / K);
            const long q = pairs_index(j, rcf_kpos((int)(i - (long)j * K), RS, Cs, kch), Cout);
            planes[q] = h;
            planes[q + 4] = (_Float16)(v - (float)h);
        }
        return;
    }
    __shared__ float tile[32][33];
    const int rs = blockIdx.z, c0 = blockIdx.x * 32, o0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const int co = o0 + r, c = c0 + tx;
        tile[r][tx] = (co < Cout && c < Cin) ? w[((long)co * RS + rs) * Cin + c] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r, co = o0 + tx;
        if (c < Cin && co < Cout) {
            const float v = tile[tx][r] * sc;
            const _Float16 h = (_Float16)v;
            const long q = pairs_index(c, rcf_kpos(rs * Cout + co, RS, Cs, kch), Cin);
            planes[q] = h;
            planes[q + 4] = (_Float16)(v - (float)h);
        }
    }
}

// pairs2[(((k >> 4) * 2 + plane) * rows + j) * 16 + (((k >> 3) & 1) ^ ((j >> 3) & 1)) * 8 + (k & 7)]: per K-step and plane the
// rows' 32-byte runs back to back, the two 16-byte halves of a run swapped on rows with bit 3 set -- byte for byte the LDS
// image conv_h2p_kernel's fragment reads expect, so a 1 KB LDS-DMA piece is 1 KB of consecutive global memory
__device__ __forceinline__ long pairs2_index(int j, int k, int rows, int plane) {
    return ((((long)(k >> 4) * 2 + plane) * rows + j) << 4) + ((((k >> 3) & 1) ^ ((j >> 3) & 1)) << 3) + (k & 7);
}

template <bool TRANSPOSE>
__global__ void __launch_bounds__(256) weight_pairs2_kernel(const float *__restrict__ w, const unsigned *__restrict__ amax,
                                                            _Float16 *__restrict__ planes, int Cout, int Cin, int RS, int korder) {
    const float sc = pow2f(h2_exponent(*amax));
    const long n = (long)Cout * RS * Cin;
    const int Cs = TRANSPOSE ? Cout : Cin, kch = rcf_kchunk(korder, RS, Cs, RCF_KCHUNK_F32);
    if (!TRANSPOSE) {
        const long step = (long)gridDim.x * blockDim.x;
        const int K = RS * Cin;
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step) {
            const float v = w[i] * sc;
            const _Float16 h = (_Float16)v;
            const int j = (int)(i / K), k = rcf_kpos((int)(i - (long)j * K), RS, Cs, kch);
            planes[pairs2_index(j, k, Cout, 0)] = h;
            planes[pairs2_index(j, k, Cout, 1)] = (_Float16)(v - (float)h);
        }
        return;
    }
    __shared__ float tile[32][33];
    const int rs = blockIdx.z, c0 = blockIdx.x * 32, o0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const int co = o0 + r, c = c0 + tx;
        tile[r][tx] = (co < Cout && c < Cin) ? w[((long)co * RS + rs) * Cin + c] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r, co = o0 + tx;
        if (c < Cin && co < Cout) {
            const float v = tile[tx][r] * sc;
            const _Float16 h = (_Float16)v;
            const int k = rcf_kpos(rs * Cout + co, RS, Cs, kch);
            planes[pairs2_index(c, k, Cin, 0)] = h;
            planes[pairs2_index(c, k, Cin, 1)] = (_Float16)(v - (float)h);
        }
    }
}

// ---- batched forms of absmax / weight_pairs / weight_pairs2 over a table of weights (rcf_common.h: rcf_wprep_entry)
__global__ void __launch_bounds__(256) wprep_absmax_kernel(const rcf_wprep_entry *__restrict__ tab, int n) {
    __shared__ unsigned sh[4];
    const rcf_wprep_entry t = tab[rcf_wprep_find(tab, n, blockIdx.x)];
    const long total4 = (long)t.Cout * t.RS * t.Cin / 4, step = (long)t.nblocks * 256;
    unsigned mx = 0u;
    for (long i = (long)(blockIdx.x - t.first_block) * 256 + threadIdx.x; i < total4; i += step) {
        const f32x4 v = reinterpret_cast<const f32x4 *>(t.w)[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) mx = max(mx, __float_as_uint(fabsf(v[e])));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = max(mx, (unsigned)__shfl_xor((int)mx, o));
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned m = max(max(sh[0], sh[1]), max(sh[2], sh[3]));
        if (m > __hip_atomic_load(t.amax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(t.amax, m);
    }
}

// out = [pairs | pairs2 (flags bit 0)] of rcf_conv_weight_pairs2_f32(transpose = 0): the same values as weight_pairs_kernel<false>
// and weight_pairs2_kernel<false>
__global__ void __launch_bounds__(256) wprep_pairs_kernel(const rcf_wprep_entry *__restrict__ tab, int n, int korder) {
    const rcf_wprep_entry t = tab[rcf_wprep_find(tab, n, blockIdx.x)];
    const float sc = pow2f(h2_exponent(*t.amax));
    const int K = t.RS * t.Cin, kch = rcf_kchunk(korder, t.RS, t.Cin, RCF_KCHUNK_F32);
    const long total = (long)t.Cout * K, step = (long)t.nblocks * 256;
    _Float16 *planes = reinterpret_cast<_Float16 *>(t.out);
    _Float16 *planes2 = planes + (long)((K + 15) / 16) * t.Cout * 32;        // pairs: cdiv(K, 16) * Cout * 64 bytes
    const bool second = t.flags & 1;
    for (long i = (long)(blockIdx.x - t.first_block) * 256 + threadIdx.x; i < total; i += step) {
        const float v = t.w[i] * sc;
        const _Float16 h = (_Float16)v, m = (_Float16)(v - (float)h);
        const int j = (int)(i / K), k = rcf_kpos((int)(i - (long)j * K), t.RS, t.Cin, kch);
        const long q = pairs_index(j, k, t.Cout);
        planes[q] = h;
        planes[q + 4] = m;
        if (second) {
            planes2[pairs2_index(j, k, t.Cout, 0)] = h;
            planes2[pairs2_index(j, k, t.Cout, 1)] = m;
        }
    }
}

// the transposed buffers (rcf_conv_weight_pairs2_f32(transpose = 1)); a block is one 32 x 32 (co, c) tile of one tap
__global__ void __launch_bounds__(256) wprep_pairs_t_kernel(const rcf_wprep_entry *__restrict__ tab, int n, int korder) {
    __shared__ float tile[32][33];
    const rcf_wprep_entry t = tab[rcf_wprep_find(tab, n, blockIdx.x)];
    const float sc = pow2f(h2_exponent(*t.amax));
    const int kch = rcf_kchunk(korder, t.RS, t.Cout, RCF_KCHUNK_F32);
    const int nbx = (t.Cin + 31) >> 5, nby = (t.Cout + 31) >> 5;
    int lb = blockIdx.x - t.first_block;
    const int rs = lb / (nbx * nby);
    lb -= rs * nbx * nby;
    const int by = lb / nbx, bx = lb - by * nbx;
    const int c0 = bx * 32, o0 = by * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const int co = o0 + r, c = c0 + tx;
        tile[r][tx] = (co < t.Cout && c < t.Cin) ? t.w[((long)co * t.RS + rs) * t.Cin + c] : 0.f;
    }
    __syncthreads();
    _Float16 *planes = reinterpret_cast<_Float16 *>(t.out);
    _Float16 *planes2 = planes + (long)((t.RS * t.Cout + 15) / 16) * 16 * t.Cin * 2;       // pairs_t: cdiv(RS Cout, 16) * 16 * Cin * 4 bytes
    const bool second = t.flags & 1;
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r, co = o0 + tx;
        if (c < t.Cin && co < t.Cout) {
            const float v = tile[tx][r] * sc;
            const _Float16 h = (_Float16)v, m = (_Float16)(v - (float)h);
            const int k = rcf_kpos(rs * t.Cout + co, t.RS, t.Cout, kch);
            const long q = pairs_index(c, k, t.Cin);
            planes[q] = h;
            planes[q + 4] = m;
            if (second) {
                planes2[pairs2_index(c, k, t.Cin, 0)] = h;
                planes2[pairs2_index(c, k, t.Cin, 1)] = m;
            }
        }
    }
}

#include "igemm_h2p.inc"
#include "igemm_h2d.inc"

// ------------------------------------------------------------------------------------------- wgrad
struct WgradParams {
    const float *X, *DY;
    float *OUT;
    int Cout, Cin, R, S;
    int H, W, Ho, Wo, stride, pad, dil;
    int x_pitch, dy_pitch;
    long M;            // N*Ho*Wo
    long chunk;        // pixels per K-split (multiple of BK)
    int itiles, jtiles;
    long split_stride; // Cout*R*S*Cin
    int beta;          // only honoured when gridDim.z == 1
    int ry0, rx0, rh, rw;   // split-bf16 kernel: contributing output pixels = this rectangle of every image (M = N*rr)
    int rband, rr;          // frame thickness (0 = whole rectangle), pixels per image
    const unsigned *amax_a, *amax_b;   // fp16-pair kernels: max |dy|, max |x| (raw fp32 bits, device scalars)
    int xcd_map;       // igemm_wgrad_h2t_kernel: rcf_wgrad_item mode (1 = an XCD's workgroups share their pixels)
    int cblocks;       // igemm_wgrad_h2t_kernel: rcf_wgrad_tile_ij (> 0: column tiles per tap, tiles numbered channel-block-major)
};

// dw[co][rs][c] = sum_m dy[m][co] * x[src(m, rs)][c].  rows i = co, cols j = c, K = pixels.
// SMALLC (Cin == 4, e.g. the zero-padded RGB stem): the taps become GEMM columns (j = rs*4 + c) instead of
// grid.y, so a 7x7x4 filter is 196 useful columns rather than 49 launches of a 94 %-empty 64-wide tile.
template <int MR, int NR, bool INCR, bool SMALLC>
__global__ void __launch_bounds__(256) igemm_wgrad_kernel(WgradParams p) {
    constexpr int BM = 64 * MR, BN = 64 * NR;
    constexpr int LDA = BM + 4, LDB = BN + 4;
    constexpr int STAGE = BK * (LDA + LDB);
    __shared__ __attribute__((aligned(16))) float smem[2 * STAGE];

    const int tile_i = blockIdx.x / p.jtiles, tile_j = blockIdx.x - tile_i * p.jtiles;
    const int i0 = tile_i * BM, j0 = tile_j * BN;
    const int rs = SMALLC ? (j0 + (int)(threadIdx.x % (BN / 4)) * 4) / 4 : (int)blockIdx.y;   // SMALLC: this thread's tap
    const int r = rs / p.S, s = rs - r * p.S;
    const long kbeg = (long)blockIdx.z * p.chunk;
    const long kend = min(p.M, kbeg + p.chunk);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    constexpr int ATPR = BM / 4, AROWS = 256 / ATPR;
    constexpr int BTPR = BN / 4, BROWS = 256 / BTPR;
    const int aiq = tid % ATPR, akr = tid / ATPR;
    const int bjq = tid % BTPR, bkr = tid / BTPR;
    const int HoWo = p.Ho * p.Wo;
    const int dy_off = (p.pad);

    f32x4 ra[MR], rb[NR];
    bool va[MR], vb[NR];
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    // pixel (n, y, x) of each gathered row, advanced by BK per K-step (no divisions in the loop)
    int pn[NR], py[NR], px_[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const long m = kbeg + bkr + i * BROWS;
        pn[i] = (int)(m / HoWo);
        const int pix = (int)(m - (long)pn[i] * HoWo);
        py[i] = pix / p.Wo;
        px_[i] = pix - py[i] * p.Wo;
    }
    const int co = i0 + aiq * 4, cc = j0 + bjq * 4;
    const int ncols = SMALLC ? p.R * p.S * 4 : p.Cin;     // useful GEMM columns
    const int xch = SMALLC ? 0 : cc;                      // channel offset inside the gathered pixel
    const bool cov = co < p.Cout, ccv = cc < ncols;
    // branch-free: invalid rows read the buffer base and are masked to zero
    auto load_tile = [&](long kt) {
#pragma unroll
        for (int i = 0; i < MR; ++i) {
            const long m = kbeg + kt * BK + akr + i * AROWS;
            const bool v = m < kend && cov;
            ra[i] = *reinterpret_cast<const f32x4 *>(v ? p.DY + m * p.dy_pitch + co : p.DY);
            va[i] = v;
        }
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const long m = kbeg + kt * BK + bkr + i * BROWS;
            const int sy = py[i] * p.stride - dy_off + r * p.dil;
            const int sx = px_[i] * p.stride - dy_off + s * p.dil;
            const bool v = m < kend && ccv && (unsigned)sy < (unsigned)p.H && (unsigned)sx < (unsigned)p.W;
            rb[i] = *reinterpret_cast<const f32x4 *>(
                v ? p.X + (((long)pn[i] * p.H + sy) * p.W + sx) * p.x_pitch + xch : p.X);
            vb[i] = v;
            // advance to the row this thread gathers in the next K-step
            if (INCR) {                       // Wo >= BK: at most one row wrap per K-step, as selects
                px_[i] += BK;
                const bool wx = px_[i] >= p.Wo;
                px_[i] -= wx ? p.Wo : 0;
                py[i] += wx ? 1 : 0;
                const bool wy = py[i] == p.Ho;
                py[i] = wy ? 0 : py[i];
                pn[i] += wy ? 1 : 0;
            } else {
                const long mn = m + BK;
                pn[i] = (int)(mn / HoWo);
                const int pix = (int)(mn - (long)pn[i] * HoWo);
                py[i] = pix / p.Wo;
                px_[i] = pix - py[i] * p.Wo;
            }
        }
    };
    auto store_tile = [&](int buf) {
        float *As = smem + buf * STAGE;
        float *Bs = As + BK * LDA;
#pragma unroll
        for (int i = 0; i < MR; ++i)
            *reinterpret_cast<f32x4 *>(As + (akr + i * AROWS) * LDA + aiq * 4) = va[i] ? ra[i] : zero4;
#pragma unroll
        for (int i = 0; i < NR; ++i)
            *reinterpret_cast<f32x4 *>(Bs + (bkr + i * BROWS) * LDB + bjq * 4) = vb[i] ? rb[i] : zero4;
    };

    f32x16 acc[MR][NR];
#pragma unroll
    for (int mr = 0; mr < MR; ++mr)
#pragma unroll
        for (int nr = 0; nr < NR; ++nr)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mr][nr][e] = 0.f;

    // the launch plan guarantees every split owns at least one pixel (kend > kbeg)
    const long KT = (kend - kbeg + BK - 1) / BK;
    load_tile(0);
    store_tile(0);
    __syncthreads();
    for (long kt = 0; kt + 1 < KT; ++kt) {
        const int cur = (int)(kt & 1);
        load_tile(kt + 1);
        __builtin_amdgcn_sched_barrier(0);      // loads stay above the MFMAs (see igemm_conv_kernel)
        const float *As = smem + cur * STAGE;
        mma_tile<MR, NR, BK>(As, As + BK * LDA, LDA, LDB, wm, wn, lane, acc);
        __builtin_amdgcn_sched_barrier(0);
        store_tile(cur ^ 1);
        __syncthreads();
    }
    {
        const float *As = smem + (int)((KT - 1) & 1) * STAGE;
        mma_tile<MR, NR, BK>(As, As + BK * LDA, LDA, LDB, wm, wn, lane, acc);
    }

    float *out = p.OUT + (long)blockIdx.z * p.split_stride;
    const long row_pitch = (long)p.R * p.S * p.Cin;
    const int l31 = lane & 31, kh = lane >> 5;
    const int rs_out = SMALLC ? 0 : (int)blockIdx.y;
#pragma unroll
    for (int nr = 0; nr < NR; ++nr) {
        const int c = j0 + wn * 32 * NR + nr * 32 + l31;
        if (c >= ncols) continue;
#pragma unroll
        for (int mr = 0; mr < MR; ++mr) {
            const int rbase = i0 + wm * 32 * MR + mr * 32 + 4 * kh;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int co = rbase + (e & 3) + 8 * (e >> 2);
                if (co < p.Cout) {
                    float *dst = out + co * row_pitch + (long)rs_out * p.Cin + c;
                    float v = acc[mr][nr][e];
                    if (p.beta && gridDim.z == 1) v += *dst;
                    *dst = v;
                }
            }
        }
    }
}

// Weight gradient on the split-bf16 path.  K = pixels is the strided dimension of both operands (dy and
// x are channel-contiguous), so the loader transposes in registers: a thread loads one channel quad of
// FOUR consecutive pixels (4 x float4), and for each of its 4 channels splits the 4 pixel values into
// bf16 planes and writes them k-contiguous (one ds_write_b64 per plane).  Waves 0-1 stage dy (rows = co),
// waves 2-3 gather x at the block's tap (rows = c).  Tile rows live at x3_prow(row) so that the 16 lanes
// of a ds_write_b64 group (4 pixel groups x 4 channel quads) fill one aligned 128-byte window.
template <int MR, int NR, int NP = 3>
__global__ void __launch_bounds__(256) igemm_wgrad_x3_kernel(WgradParams p) {
    constexpr int BM = 64 * MR, BN = 64 * NR;
    constexpr int PA = BM * 32, PB = BN * 32, STAGE = NP * (PA + PB);
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];

    const int tile_i = blockIdx.x / p.jtiles, tile_j = blockIdx.x - tile_i * p.jtiles;
    const int i0 = tile_i * BM, j0 = tile_j * BN;
    const int rs = (int)blockIdx.y;
    const int r = rs / p.S, s = rs - r * p.S;
    const long kbeg = (long)blockIdx.z * p.chunk;
    const long kend = min(p.M, kbeg + p.chunk);
    const int klen = (int)(kend - kbeg);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const bool isB = tid >= 128;                 // wave-uniform role
    const int t = tid & 127, g = t & 3, q = t >> 2;
    const int HoWo = p.rr;                        // contributing pixels per image
    const int n_first = (int)(kbeg / HoWo);
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(p.DY + (long)n_first * p.Ho * p.Wo * p.dy_pitch), 0, (int)X3_OOB, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(p.X + (long)n_first * p.H * p.W * p.x_pitch), 0, (int)X3_OOB, 0x00020000);

    const int ch = (isB ? j0 : i0) + 4 * q;
    const bool active = isB ? (q < BN / 4 && ch < p.Cin) : (q < BM / 4 && ch < p.Cout);
    // (image, row, column) of this thread's 4 pixels -- image coordinates -- advanced by BK per K-step
    int pn[4], py[4], px_[4], ppix[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const long m = kbeg + 4 * g + i;
        pn[i] = (int)(m / HoWo);
        ppix[i] = (int)(m - (long)pn[i] * HoWo);          // pixel index inside the image's region
        region_yx(ppix[i], p.ry0, p.rx0, p.rh, p.rw, p.rband, py[i], px_[i]);
        pn[i] -= n_first;
    }
    const int row_l = 4 * q;                                              // first tile row of this thread
    const int prow0 = x3_prow(row_l);                                     // rows row_l + e -> prow0 + 4 e
    const int st_off = prow0 * 32 + (g & 1) * 8;                         // bit 3 of prow0 is clear; row e sets it to e>>1
    const bool incr = p.rw >= BK && p.rband <= 0;          // rows of a rectangle at least one K-step wide: walk by increments

    f32x4 rr[4];
    auto load_tile = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int mk = kt * BK + 4 * g + i;
            const int oy = py[i], ox = px_[i];                              // output pixel in the image
            if (!isB) {
                const bool v = active && mk < klen;
                const unsigned bo = v ? (unsigned)(((pn[i] * p.Ho + oy) * p.Wo + ox) * p.dy_pitch + ch) * 4u : X3_OOB;
                rr[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, (int)bo, 0, 0));
            } else {
                const int sy = oy * p.stride - p.pad + r * p.dil;
                const int sx = ox * p.stride - p.pad + s * p.dil;
                const bool v = active && mk < klen && (unsigned)sy < (unsigned)p.H && (unsigned)sx < (unsigned)p.W;
                const unsigned bo = v ? (unsigned)(((pn[i] * p.H + sy) * p.W + sx) * p.x_pitch + ch) * 4u : X3_OOB;
                rr[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB, (int)bo, 0, 0));
            }
            if (incr) {                          // rw >= BK: at most one row wrap per K-step
                px_[i] += BK;
                const bool wx = px_[i] >= p.rx0 + p.rw;
                px_[i] -= wx ? p.rw : 0;
                py[i] += wx ? 1 : 0;
                const bool wy = py[i] == p.ry0 + p.rh;
                py[i] = wy ? p.ry0 : py[i];
                pn[i] += wy ? 1 : 0;
            } else {                             // narrow rectangle / frame: advance the region-linear index
                ppix[i] += BK;
                while (ppix[i] >= HoWo) { ppix[i] -= HoWo; ++pn[i]; }
                region_yx(ppix[i], p.ry0, p.rx0, p.rh, p.rw, p.rband, py[i], px_[i]);
            }
        }
    };
    int ka = 0, kb = 0;                      // fp16 pairs: scales 2^ka of dy, 2^kb of x
    if constexpr (NP == 2) {
        ka = h2_exponent(*p.amax_a);
        kb = h2_exponent(*p.amax_b);
    }
    const float sc = pow2f(isB ? kb : ka);
    auto store_tile = [&](int buf) {
        char *base = smem + buf * STAGE + (isB ? NP * PA : 0) + st_off;
        const int plane = isB ? PB : PA;
        if (q < (isB ? BN : BM) / 4) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                u32x2 h, m, l;
                char *d = base + e * 128 + (((g >> 1) ^ (e >> 1)) << 4);  // row prow0 + 4 e, 16-byte halves swizzled
                if constexpr (NP == 2) {
                    split2h(f32x4{rr[0][e], rr[1][e], rr[2][e], rr[3][e]}, sc, h, m);
                } else {
                    split3(f32x4{rr[0][e], rr[1][e], rr[2][e], rr[3][e]}, h, m, l);
                    *reinterpret_cast<u32x2 *>(d + 2 * plane) = l;
                }
                *reinterpret_cast<u32x2 *>(d) = h;
                *reinterpret_cast<u32x2 *>(d + plane) = m;
            }
        }
    };

    f32x16 acc[MR][NR];
#pragma unroll
    for (int mr = 0; mr < MR; ++mr)
#pragma unroll
        for (int nr = 0; nr < NR; ++nr)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mr][nr][e] = 0.f;

    const int arow0 = wm * 32 * MR, brow0 = wn * 32 * NR;
    const int KT = (klen + BK - 1) / BK;
    load_tile(0);
    store_tile(0);
    __syncthreads();
    for (int kt = 0; kt + 1 < KT; ++kt) {
        const int cur = kt & 1;
        load_tile(kt + 1);
        __builtin_amdgcn_sched_barrier(0);
        const char *As = smem + cur * STAGE;
        mma_x3<MR, NR, PA, PB, true, NP>(As, As + NP * PA, arow0, brow0, lane, acc);
        store_tile(cur ^ 1);
        __syncthreads();
    }
    {
        const char *As = smem + ((KT - 1) & 1) * STAGE;
        mma_x3<MR, NR, PA, PB, true, NP>(As, As + NP * PA, arow0, brow0, lane, acc);
    }

    const float inv_a = pow2f(-ka), inv_b = pow2f(-kb);
    float *out = p.OUT + (long)blockIdx.z * p.split_stride;
    const long row_pitch = (long)p.R * p.S * p.Cin;
    const int l31 = lane & 31, kh = lane >> 5;
#pragma unroll
    for (int nr = 0; nr < NR; ++nr) {
        const int c = j0 + brow0 + nr * 32 + l31;
        if (c >= p.Cin) continue;
#pragma unroll
        for (int mr = 0; mr < MR; ++mr) {
            const int rbase = i0 + arow0 + mr * 32 + 4 * kh;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int co = rbase + (e & 3) + 8 * (e >> 2);
                if (co < p.Cout) {
                    float *dst = out + co * row_pitch + (long)rs * p.Cin + c;
                    float v = acc[mr][nr][e];
                    if constexpr (NP == 2) v = (v * inv_a) * inv_b;
                    if (p.beta && gridDim.z == 1) v += *dst;
                    *dst = v;
                }
            }
        }
    }
}

// two pixels of one channel -> three bf16 pairs (one dword per plane)
__device__ __forceinline__ void split3_pair(float a, float b, unsigned &h, unsigned &m, unsigned &l) {
    const f32x2 v = {a, b};
    const bf16x2 hv = __builtin_convertvector(v, bf16x2);
    const f32x2 r = v - __builtin_convertvector(hv, f32x2);
    const bf16x2 mv = __builtin_convertvector(r, bf16x2);
    const f32x2 q = r - __builtin_convertvector(mv, f32x2);
    const bf16x2 lv = __builtin_convertvector(q, bf16x2);
    h = __builtin_bit_cast(unsigned, hv);
    m = __builtin_bit_cast(unsigned, mv);
    l = __builtin_bit_cast(unsigned, lv);
}

__device__ __forceinline__ void split2h_pair(float a, float b, float s, unsigned &h, unsigned &m) {
    const f32x2 v = {a * s, b * s};
    const f16x2 hv = __builtin_convertvector(v, f16x2);
    const f32x2 r = v - __builtin_convertvector(hv, f32x2);
    const f16x2 mv = __builtin_convertvector(r, f16x2);
    h = __builtin_bit_cast(unsigned, hv);
    m = __builtin_bit_cast(unsigned, mv);
}

// 128 x 256 tile of the weight gradient (Cout >= 128, Cin >= 256, whole tensors only): every thread stages one
// x item (channel quad x 4 pixels, as in igemm_wgrad_x3_kernel) AND half a dy item (channel quad x 2 pixels, one
// ds_write_b32 per plane), so the loader work is balanced over the four waves; each wave owns 2 x 4 MFMA tiles.
template <int NP = 3>
__global__ void __launch_bounds__(256, 2) igemm_wgrad_x3_wide_kernel(WgradParams p) {
    constexpr int MR = 2, NR = 4, BM = 128, BN = 256;
    constexpr int PA = BM * 32, PB = BN * 32, STAGE = NP * (PA + PB);
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];

    const int tile_i = blockIdx.x / p.jtiles, tile_j = blockIdx.x - tile_i * p.jtiles;
    const int i0 = tile_i * BM, j0 = tile_j * BN;
    const int rs = (int)blockIdx.y;
    const int r = rs / p.S, s = rs - r * p.S;
    const long kbeg = (long)blockIdx.z * p.chunk;
    const long kend = min(p.M, kbeg + p.chunk);
    const int klen = (int)(kend - kbeg);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int g = tid & 3, q = tid >> 2;          // x item: pixels 4g..4g+3, channel quad q (0..63)
    const int g2 = tid & 7, qa = tid >> 3;        // dy half-item: pixels 2 g2, 2 g2 + 1, channel quad qa (0..31)
    const int HoWo = p.Ho * p.Wo;
    const int n_first = (int)(kbeg / HoWo);
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(p.DY + kbeg * p.dy_pitch), 0, (int)X3_OOB, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(p.X + (long)n_first * p.H * p.W * p.x_pitch), 0, (int)X3_OOB, 0x00020000);

    const int cha = i0 + 4 * qa, chb = j0 + 4 * q;
    const bool acta = cha < p.Cout, actb = chb < p.Cin;
    int pn[4], py[4], px_[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const long m = kbeg + 4 * g + i;
        pn[i] = (int)(m / HoWo);
        const int pix = (int)(m - (long)pn[i] * HoWo);
        py[i] = pix / p.Wo;
        px_[i] = pix - py[i] * p.Wo;
        pn[i] -= n_first;
    }
    const int prow_b = x3_prow(4 * q), prow_a = x3_prow(4 * qa);
    const int st_b = prow_b * 32 + (g & 1) * 8;
    const int st_a = prow_a * 32 + (g2 & 3) * 4;
    const bool incr = p.Wo >= BK;

    f32x4 ra[2], rb[4];
    auto load_tile = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int mk = kt * BK + 2 * g2 + i;
            const unsigned bo = (acta && mk < klen) ? (unsigned)(mk * p.dy_pitch + cha) * 4u : X3_OOB;
            ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, (int)bo, 0, 0));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int mk = kt * BK + 4 * g + i;
            const int sy = py[i] * p.stride - p.pad + r * p.dil;
            const int sx = px_[i] * p.stride - p.pad + s * p.dil;
            const bool v = actb && mk < klen && (unsigned)sy < (unsigned)p.H && (unsigned)sx < (unsigned)p.W;
            const unsigned bo = v ? (unsigned)(((pn[i] * p.H + sy) * p.W + sx) * p.x_pitch + chb) * 4u : X3_OOB;
            rb[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB, (int)bo, 0, 0));
            if (incr) {
                px_[i] += BK;
                const bool wx = px_[i] >= p.Wo;
                px_[i] -= wx ? p.Wo : 0;
                py[i] += wx ? 1 : 0;
                const bool wy = py[i] == p.Ho;
                py[i] = wy ? 0 : py[i];
                pn[i] += wy ? 1 : 0;
            } else {
                const long mn = kbeg + (long)(kt + 1) * BK + 4 * g + i;
                const int nn = (int)(mn / HoWo);
                const int pix = (int)(mn - (long)nn * HoWo);
                pn[i] = nn - n_first;
                py[i] = pix / p.Wo;
                px_[i] = pix - py[i] * p.Wo;
            }
        }
    };
    int ka = 0, kb = 0;                      // fp16 pairs: scales 2^ka of dy, 2^kb of x
    if constexpr (NP == 2) {
        ka = h2_exponent(*p.amax_a);
        kb = h2_exponent(*p.amax_b);
    }
    const float sa = pow2f(ka), sb = pow2f(kb);
    auto store_tile = [&](int buf) {
        char *As = smem + buf * STAGE, *Bs = As + NP * PA;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            unsigned h, m, l;
            char *d = As + st_a + e * 128 + (((g2 >> 2) ^ (e >> 1)) << 4);     // row prow_a + 4 e
            if constexpr (NP == 2) {
                split2h_pair(ra[0][e], ra[1][e], sa, h, m);
            } else {
                split3_pair(ra[0][e], ra[1][e], h, m, l);
                *reinterpret_cast<unsigned *>(d + 2 * PA) = l;
            }
            *reinterpret_cast<unsigned *>(d) = h;
            *reinterpret_cast<unsigned *>(d + PA) = m;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            u32x2 h, m, l;
            char *d = Bs + st_b + e * 128 + (((g >> 1) ^ (e >> 1)) << 4);      // row prow_b + 4 e
            if constexpr (NP == 2) {
                split2h(f32x4{rb[0][e], rb[1][e], rb[2][e], rb[3][e]}, sb, h, m);
            } else {
                split3(f32x4{rb[0][e], rb[1][e], rb[2][e], rb[3][e]}, h, m, l);
                *reinterpret_cast<u32x2 *>(d + 2 * PB) = l;
            }
            *reinterpret_cast<u32x2 *>(d) = h;
            *reinterpret_cast<u32x2 *>(d + PB) = m;
        }
    };

    f32x16 acc[MR][NR];
#pragma unroll
    for (int mr = 0; mr < MR; ++mr)
#pragma unroll
        for (int nr = 0; nr < NR; ++nr)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mr][nr][e] = 0.f;

    const int arow0 = wm * 32 * MR, brow0 = wn * 32 * NR;
    const int KT = (klen + BK - 1) / BK;
    load_tile(0);
    store_tile(0);
    __syncthreads();
    for (int kt = 0; kt + 1 < KT; ++kt) {
        const int cur = kt & 1;
        load_tile(kt + 1);
        __builtin_amdgcn_sched_barrier(0);
        const char *As = smem + cur * STAGE;
        mma_x3<MR, NR, PA, PB, true, NP>(As, As + NP * PA, arow0, brow0, lane, acc);
        store_tile(cur ^ 1);
        __syncthreads();
    }
    {
        const char *As = smem + ((KT - 1) & 1) * STAGE;
        mma_x3<MR, NR, PA, PB, true, NP>(As, As + NP * PA, arow0, brow0, lane, acc);
    }

    const float inv_a = pow2f(-ka), inv_b = pow2f(-kb);
    float *out = p.OUT + (long)blockIdx.z * p.split_stride;
    const long row_pitch = (long)p.R * p.S * p.Cin;
    const int l31 = lane & 31, kh = lane >> 5;
#pragma unroll
    for (int mr = 0; mr < MR; ++mr) {
        const int rbase = i0 + arow0 + mr * 32 + 4 * kh;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int co = rbase + (e & 3) + 8 * (e >> 2);
            if (co >= p.Cout) continue;
            float *drow = out + co * row_pitch + (long)rs * p.Cin + j0 + brow0 + l31;
#pragma unroll
            for (int nr = 0; nr < NR; ++nr) {
                if (j0 + brow0 + nr * 32 + l31 >= p.Cin) continue;
                float v = acc[mr][nr][e];
                if constexpr (NP == 2) v = (v * inv_a) * inv_b;
                if (p.beta && gridDim.z == 1) v += drow[nr * 32];
                drow[nr * 32] = v;
            }
        }
    }
}

// The same 128 x 256 weight-gradient tile with the operands staged in their NATURAL order: LDS holds, per fp16 plane,
// [16 pixels][channels] (a float4 = 4 channels of a pixel is split once and written with one ds_write_b64 per plane;
// the global loads of a K-step are whole runs of 512 B / 1 KB per pixel), and the MFMA's k-contiguous fragments come out
// of gfx950's transposing LDS read: ds_read_b64_tr_b16 gives lane c of a 16-lane group the 4 k-values of channel
// c0 + c when lane p of the group points at row k0 + p/4, channels c0 + 4 (p%4) .. +3.  No register transposes, no
// row permutation.  Row pitch = channels * 2 + 64 bytes: the 64-byte runs that a 32-lane half reads from 4 rows tile the
// 256-byte bank space (a pitch of 32 mod 256 made neighbouring rows overlap by half: SQ_LDS_BANK_CONFLICT 33 % of the
// LDS cycles).
typedef short s16x4 __attribute__((__vector_size__(4 * sizeof(short))));
__device__ __forceinline__ u32x2 lds_tr16(const char *p) {
    const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4 *)(__attribute__((address_space(3))) char *)p);
    return __builtin_bit_cast(u32x2, v);
}

// NR = 4: 128 x 256 tile; NR = 2: 128 x 128 (operands too narrow for the wide tile).
// The GEMM columns are (tap, input channel) pairs, j = rs * Cin + c -- the weight's own memory order -- and not grid rows:
// a tile of a narrow layer (Cin = 64: 576 columns of a 3x3 filter) spans several taps, so dy is read once per 256 columns
// instead of once per tap.  Cin % 64 == 0: each of a thread's 64-channel groups lies in ONE tap, which is block-uniform
// (scalar registers).  ONETAP: Cin % BN == 0, the whole tile belongs to one tap.
// REGION: only the pixels of p's rectangle / frame contribute (M = N * rr); the pixel walk divides.
// MR = 4 (rcf_conv_set_wgrad_big): a 256 x 256 tile, ONE workgroup per CU -- 128 x 128 accumulators per wave (256 registers),
// 74 KB of LDS: a third fewer loads, splits, LDS writes and fragment reads per MFMA than the 128 x 256 tile.
template <int NR, bool REGION, bool ONETAP, int MR = 2>
__global__ void __launch_bounds__(256, MR == 2 ? 2 : 1) igemm_wgrad_h2t_kernel(WgradParams p) {
    constexpr int BM = 64 * MR, BN = 64 * NR;
    constexpr int NAP = BM / 64, NBP = BN / 64;                  // 64-channel groups per thread: dy, x
    constexpr int PIA = BM * 2 + 64, PIB = BN * 2 + 64;          // row (pixel) pitch of the dy / x planes, bytes (= 64 mod 256)
    constexpr int PLA = BK * PIA, PLB = BK * PIB, STAGE = 2 * (PLA + PLB);
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];

    int tile, split;
    rcf_wgrad_item(p.xcd_map, tile, split);
    int tile_i, tile_j;
    rcf_wgrad_tile_ij(tile, p.itiles, p.jtiles, p.cblocks, tile_i, tile_j);
    const int i0 = tile_i * BM, j0 = tile_j * BN;
    const int Ktot = p.R * p.S * p.Cin;
    const long kbeg = (long)split * p.chunk;
    const long kend = min(p.M, kbeg + p.chunk);
    const int klen = (int)(kend - kbeg);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int HoWo = p.rr;                                    // contributing pixels per image
    const int n_first = (int)(kbeg / HoWo);
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(p.DY + (REGION ? (long)n_first * p.Ho * p.Wo : kbeg) * p.dy_pitch), 0, (int)X3_OOB, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(p.X + (long)n_first * p.H * p.W * p.x_pitch), 0, (int)X3_OOB, 0x00020000);

    // loader roles: thread t serves ONE pixel (t >> 4) of the K-step in both operands -- channel quads (t & 15) + 16 j of dy
    // (j < NAP) and of x (j < NBP): one pixel walk and one bounds test per K-step (the kernel used to spend as many VALU
    // instructions on addresses as on splitting); 16 lanes x 16 B = 256 contiguous bytes per pixel and instruction
    const int q0 = tid & 15, pb0 = tid >> 4;
    const int cha = i0 + 4 * q0;
    // block-uniform taps of the 64-column groups (scalar)
    int tdy[NBP], tdx[NBP], cb[NBP];
#pragma unroll
    for (int j = 0; j < NBP; ++j) {
        const int jc = j0 + (ONETAP ? 0 : 64 * j);
        const int rs = jc / p.Cin;
        const int r = rs / p.S;
        tdy[j] = r * p.dil - p.pad;
        tdx[j] = (rs - r * p.S) * p.dil - p.pad;
        cb[j] = j0 + 64 * j - rs * p.Cin;                     // first channel of the group inside its tap
    }
    // pixel walk: the position of the thread's pixel of the current K-step, advanced by BK per step (Wo >= BK: at most one
    // row wrap).  Narrow images and regions take the division path.
    int pn, py, px_;
    {
        const long m = kbeg + pb0;
        pn = (int)(m / HoWo);
        const int pix = (int)(m - (long)pn * HoWo);
        py = pix / p.Wo;
        px_ = pix - py * p.Wo;
        pn -= n_first;
    }
    const bool incr = !REGION && p.Wo >= BK;

    // both operands come from HBM: their loads run TWO K-steps ahead (two register sets, ping-pong by the parity of the step)
    f32x4 ra[2][NAP], rb[2][NBP];
    auto load_tile = [&](auto INCR, int kt, f32x4 (&ra)[NAP], f32x4 (&rb)[NBP]) {
        const int mk = kt * BK + pb0;
        int n, y, x;
        if constexpr (decltype(INCR)::value) {
            n = pn; y = py; x = px_;
        } else {
            const long m = kbeg + mk;
            const int nn = (int)(m / HoWo);
            const int pix = (int)(m - (long)nn * HoWo);
            n = nn - n_first;
            if constexpr (REGION) {
                region_yx(pix, p.ry0, p.rx0, p.rh, p.rw, p.rband, y, x);
            } else {
                y = pix / p.Wo;
                x = pix - y * p.Wo;
            }
        }
        const int inb = (int)(mk < klen);
        const int dyoff = (REGION ? ((n * p.Ho + y) * p.Wo + x) * p.dy_pitch : mk * p.dy_pitch) + cha;
#pragma unroll
        for (int j = 0; j < NAP; ++j) {
            const unsigned bo = x3_oob_unless((unsigned)(dyoff + 64 * j) * 4u, inb & (int)(cha + 64 * j < p.Cout));
            ra[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, (int)bo, 0, 0));
        }
        const int ys = y * p.stride, xs = x * p.stride;
        if constexpr (ONETAP) {
            const int sy = ys + tdy[0], sx = xs + tdx[0];
            const int v = inb & (int)((unsigned)sy < (unsigned)p.H) & (int)((unsigned)sx < (unsigned)p.W);
            const int xoff = ((n * p.H + sy) * p.W + sx) * p.x_pitch + cb[0] + 4 * q0;
#pragma unroll
            for (int j = 0; j < NBP; ++j) {
                const unsigned bo = x3_oob_unless((unsigned)(xoff + 64 * j) * 4u, v & (int)(j0 + 64 * j + 4 * q0 < Ktot));
                rb[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB, (int)bo, 0, 0));
            }
        } else {
#pragma unroll
            for (int j = 0; j < NBP; ++j) {
                const int sy = ys + tdy[j], sx = xs + tdx[j];
                const int v = inb & (int)((unsigned)sy < (unsigned)p.H) & (int)((unsigned)sx < (unsigned)p.W) &
                              (int)(j0 + 64 * j + 4 * q0 < Ktot);
                const unsigned bo = x3_oob_unless((unsigned)(((n * p.H + sy) * p.W + sx) * p.x_pitch + cb[j] + 4 * q0) * 4u, v);
                rb[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB, (int)bo, 0, 0));
            }
        }
        if constexpr (decltype(INCR)::value) {       // K-steps are loaded in order: advance the pixel by BK
            px_ += BK;
            const bool wx = px_ >= p.Wo;
            px_ -= wx ? p.Wo : 0;
            py += wx ? 1 : 0;
            const bool wy = py == p.Ho;
            py = wy ? 0 : py;
            pn += wy ? 1 : 0;
        }
    };
    const int ka = h2_exponent(*p.amax_a), kb = h2_exponent(*p.amax_b);
    const float sa = pow2f(ka), sb = pow2f(kb);
    auto store_tile = [&](int buf, const f32x4 (&ra)[NAP], const f32x4 (&rb)[NBP]) {
        char *As = smem + buf * STAGE, *Bs = As + 2 * PLA;
#pragma unroll
        for (int j = 0; j < NAP; ++j) {
            u32x2 h, m;
            split2h(ra[j], sa, h, m);
            char *d = As + pb0 * PIA + (q0 + 16 * j) * 8;
            *reinterpret_cast<u32x2 *>(d) = h;
            *reinterpret_cast<u32x2 *>(d + PLA) = m;
        }
#pragma unroll
        for (int j = 0; j < NBP; ++j) {
            u32x2 h, m;
            split2h(rb[j], sb, h, m);
            char *d = Bs + pb0 * PIB + (q0 + 16 * j) * 8;
            *reinterpret_cast<u32x2 *>(d) = h;
            *reinterpret_cast<u32x2 *>(d + PLB) = m;
        }
    };

    f32x16 acc[MR][NR];
#pragma unroll
    for (int mr = 0; mr < MR; ++mr)
#pragma unroll
        for (int nr = 0; nr < NR; ++nr)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mr][nr][e] = 0.f;

    const int arow0 = wm * 32 * MR, brow0 = wn * 32 * NR;
    // fragment addressing: 16-lane group gi = lane >> 4 serves rows 16 (gi & 1) .. +15 of a 32-row tile and the k half
    // gi >> 1; inside the group lane q points at pixel row 8 (gi >> 1) + q / 4 (+ 4 for the second read), channels 4 (q % 4)
    const int gi = lane >> 4, q16 = lane & 15;
    const int fa = (8 * (gi >> 1) + (q16 >> 2)) * PIA + (arow0 + 16 * (gi & 1) + 4 * (q16 & 3)) * 2;
    const int fb = (8 * (gi >> 1) + (q16 >> 2)) * PIB + (brow0 + 16 * (gi & 1) + 4 * (q16 & 3)) * 2;
    auto mma = [&](int buf) {
        const char *As = smem + buf * STAGE, *Bs = As + 2 * PLA;
        f16x8 a[MR][2], b[NR][2];
#pragma unroll
        for (int mr = 0; mr < MR; ++mr)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
                const u32x2 lo = lds_tr16(As + pl * PLA + fa + mr * 64);
                const u32x2 hi = lds_tr16(As + pl * PLA + fa + mr * 64 + 4 * PIA);
                a[mr][pl] = __builtin_bit_cast(f16x8, u32x4{lo[0], lo[1], hi[0], hi[1]});
            }
#pragma unroll
        for (int nr = 0; nr < NR; ++nr)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
                const u32x2 lo = lds_tr16(Bs + pl * PLB + fb + nr * 64);
                const u32x2 hi = lds_tr16(Bs + pl * PLB + fb + nr * 64 + 4 * PIB);
                b[nr][pl] = __builtin_bit_cast(f16x8, u32x4{lo[0], lo[1], hi[0], hi[1]});
            }
        constexpr int HA[3] = {1, 0, 0}, HB[3] = {0, 1, 0};
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int mr = 0; mr < MR; ++mr)
#pragma unroll
                for (int nr = 0; nr < NR; ++nr)
                    acc[mr][nr] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[mr][HA[t]], b[nr][HB[t]], acc[mr][nr], 0, 0, 0);
    };

    const int KT = (klen + BK - 1) / BK;
    // K-step kt: MFMAs on stage kt & 1 while the tiles of steps kt + 1 (loaded a step ago) and kt + 2 are in flight; then
    // tile kt + 1 is split into the other stage
    auto k_loop = [&](auto INCR) {
        load_tile(INCR, 0, ra[0], rb[0]);
        load_tile(INCR, 1, ra[1], rb[1]);    // past the end of the chunk: out-of-range offsets, zeros
        store_tile(0, ra[0], rb[0]);
        __syncthreads();
        int kt = 0;
        for (; kt + 2 < KT; kt += 2) {
            load_tile(INCR, kt + 2, ra[0], rb[0]);
            __builtin_amdgcn_sched_barrier(0);
            mma(0);
            store_tile(1, ra[1], rb[1]);
            __syncthreads();
            load_tile(INCR, kt + 3, ra[1], rb[1]);
            __builtin_amdgcn_sched_barrier(0);
            mma(1);
            store_tile(0, ra[0], rb[0]);
            __syncthreads();
        }
        mma(0);
        if (kt + 1 < KT) {
            store_tile(1, ra[1], rb[1]);
            __syncthreads();
            mma(1);
        }
    };
    if (incr) k_loop(std::true_type{});
    else k_loop(std::false_type{});

    float *out = p.OUT + (long)split * p.split_stride;
    const float inv_a = pow2f(-ka), inv_b = pow2f(-kb);
    const int l31 = lane & 31, kh = lane >> 5;
    if (i0 + BM <= p.Cout && j0 + BN <= Ktot && !(p.beta && gridDim.z == 1)) {
        // the tile lies inside the weight tensor and is a split-K partial (or overwrites): scale and store, nothing to test
        // (the general loop below is ~2 300 instructions in ~250 basic blocks; a workgroup's K-loop is ~10 000)
#pragma unroll
        for (int mr = 0; mr < MR; ++mr)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float *drow = out + (long)(i0 + arow0 + mr * 32 + 4 * kh + (e & 3) + 8 * (e >> 2)) * Ktot + j0 + brow0 + l31;
#pragma unroll
                for (int nr = 0; nr < NR; ++nr) drow[nr * 32] = (acc[mr][nr][e] * inv_a) * inv_b;
            }
        return;
    }
#pragma unroll
    for (int mr = 0; mr < MR; ++mr) {
        const int rbase = i0 + arow0 + mr * 32 + 4 * kh;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int co = rbase + (e & 3) + 8 * (e >> 2);
            if (co >= p.Cout) continue;
            float *drow = out + (long)co * Ktot + j0 + brow0 + l31;
#pragma unroll
            for (int nr = 0; nr < NR; ++nr) {
                if (j0 + brow0 + nr * 32 + l31 >= Ktot) continue;
                float v = (acc[mr][nr][e] * inv_a) * inv_b;
                if (p.beta && gridDim.z == 1) v += drow[nr * 32];
                drow[nr * 32] = v;
            }
        }
    }
}

#include "igemm_h2dw.inc"

__global__ void splitk_reduce_kernel(const float *__restrict__ ws, float *__restrict__ dw, long n4, long stride,
                                     int splits, int beta) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long step = (long)gridDim.x * blockDim.x;
    for (; i < n4; i += step) {
        // four independent chains keep four loads in flight; combined in a fixed order (deterministic)
        f32x4 a0 = beta ? reinterpret_cast<const f32x4 *>(dw)[i] : f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 a1 = {0.f, 0.f, 0.f, 0.f}, a2 = a1, a3 = a1;
        int s = 0;
        for (; s + 3 < splits; s += 4) {
            a0 += reinterpret_cast<const f32x4 *>(ws + (long)s * stride)[i];
            a1 += reinterpret_cast<const f32x4 *>(ws + (long)(s + 1) * stride)[i];
            a2 += reinterpret_cast<const f32x4 *>(ws + (long)(s + 2) * stride)[i];
            a3 += reinterpret_cast<const f32x4 *>(ws + (long)(s + 3) * stride)[i];
        }
        for (; s < splits; ++s) a0 += reinterpret_cast<const f32x4 *>(ws + (long)s * stride)[i];
        reinterpret_cast<f32x4 *>(dw)[i] = (a0 + a1) + (a2 + a3);
    }
}

int check_shape(const rcf_conv_shape *s) {
    if (!s || s->struct_bytes != sizeof(rcf_conv_shape)) return RCF_EINVAL;      // a caller built against another header
    if (s->N <= 0 || s->H <= 0 || s->W <= 0 || s->Cin <= 0 || s->Cout <= 0 || s->R <= 0 || s->S <= 0) return RCF_EINVAL;
    if (s->Cin % 4 || s->x_pitch % 4 || s->y_pitch % 4 || s->x_pitch < s->Cin || s->y_pitch < s->Cout) return RCF_EINVAL;
    if (s->stride <= 0 || s->dil <= 0 || s->pad < 0) return RCF_EINVAL;
    const int ho = (s->H + 2 * s->pad - s->dil * (s->R - 1) - 1) / s->stride + 1;
    const int wo = (s->W + 2 * s->pad - s->dil * (s->S - 1) - 1) / s->stride + 1;
    if (ho != s->Ho || wo != s->Wo) return RCF_EINVAL;
    if ((long)s->N * s->Ho * s->Wo >= (1L << 31) || (long)s->N * s->H * s->W >= (1L << 31)) return RCF_EINVAL;
    return 0;
}

// RCF_CONV_FP32_MFMA(v) in the call's flags: the fp32-MFMA kernels with tuning variant v (bit 0: K-step 32, bit 1: row-major
// LDS tiles) instead of the fp16-pair / bf16-triple kernels; -1 = not asked for
inline int fp32_mfma_variant(unsigned flags) { return (int)((flags >> 12) & 7u) - 1; }
inline bool use_x3(unsigned flags) { return fp32_mfma_variant(flags) < 0; }
// thin 1x1 convs (<= 16 output channels) leave the GEMM kernels for csrc/thin.hip's streaming passes -- on the default path only:
// the RCF_CONV_FP32_MFMA test variants keep the kernels they are there to exercise
inline bool thin_path(const rcf_conv_shape *s) { return use_x3(s->flags) && !(s->flags & RCF_CONV_NO_THIN) && rcf_thin_ok(s); }
inline bool korder_chunked(unsigned flags) { return !(flags & RCF_CONV_KORDER_NATURAL); }

inline unsigned magic_of(int d) { return d <= 1 ? 0u : (unsigned)((1ull << 32) / (unsigned)d + 1ull); }

template <int BMODE, int BKT, bool RM>
int launch_igemm_v(IgemmParams &p, hipStream_t st) {
    p.cs_magic = magic_of(p.Cs);
    p.s_magic = magic_of(p.S);
    if ((long)p.K * p.Cs >= (1L << 32)) return RCF_EINVAL;
    const bool strided = BMODE == 1 && p.div > 1;
    // tile: 128x64 for narrow outputs, 128x128 by default, 128x256 (8 accumulator tiles per wave: twice the
    // MFMA work per barrier) when the output is wide and there are enough row tiles to fill the chip
    const bool wide = p.Ncol > 64;
    const int BM = 128, BN = wide ? 128 : 64;
    p.mtiles = rcf_cdiv(p.M, BM);
    p.mtiles8 = rcf_cdiv(p.mtiles, 8);
    p.ntiles = rcf_cdiv(p.Ncol, BN);
    const int groups = rcf_cdiv(p.mtiles, 8);
    const dim3 grid((unsigned)(groups * 8 * p.ntiles));
    if (strided) {
        if (wide) hipLaunchKernelGGL((igemm_conv_kernel<2, 2, BMODE, BKT, RM, BMODE == 1>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((igemm_conv_kernel<2, 1, BMODE, BKT, RM, BMODE == 1>), grid, dim3(256), 0, st, p);
    } else {
        if (wide) hipLaunchKernelGGL((igemm_conv_kernel<2, 2, BMODE, BKT, RM, false>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((igemm_conv_kernel<2, 1, BMODE, BKT, RM, false>), grid, dim3(256), 0, st, p);
    }
    RCF_LAUNCH_CHECK();
    return 0;
}

// split-bf16 launch (B always k-contiguous)
template <int MR, int NR, int WM, int WN, int NP, bool PRE = false>
void launch_x3_cfg_np(IgemmParams &p, bool strided, hipStream_t st, int batches) {
    constexpr int BM = 32 * MR * WM, BN = 32 * NR * WN;
    p.mtiles = rcf_cdiv(p.M, BM);
    p.mtiles8 = rcf_cdiv(p.mtiles, 8);
    p.ntiles = rcf_cdiv(p.Ncol, BN);
    p.colmap = batches == 1 && p.Ncol % BN == 0 &&
               rcf_colmap_pays(!(p.flags & RCF_CONV_NO_COLMAP), (long)p.M * p.Cs * 4, (long)p.K * p.Ncol * 4, p.mtiles, p.ntiles);
    const dim3 grid((unsigned)(rcf_cdiv(p.mtiles, 8) * 8 * p.ntiles), (unsigned)batches);
    if constexpr (NP == 2 && PRE) {              // data gradient + the batch-norm backward sums (rcf_conv2d_dgrad_bnsums_f32)
        if (p.stats && p.step < 0) {
            if (strided) hipLaunchKernelGGL((igemm_conv_x3_kernel<MR, NR, WM, WN, true, true, NP, PRE, true, true>), grid, dim3(64 * WM * WN), 0, st, p);
            else hipLaunchKernelGGL((igemm_conv_x3_kernel<MR, NR, WM, WN, false, true, NP, PRE, true, true>), grid, dim3(64 * WM * WN), 0, st, p);
            return;
        }
    }
    if (strided) hipLaunchKernelGGL((igemm_conv_x3_kernel<MR, NR, WM, WN, true, true, NP, PRE>), grid, dim3(64 * WM * WN), 0, st, p);
    else if (p.step < 0) hipLaunchKernelGGL((igemm_conv_x3_kernel<MR, NR, WM, WN, false, true, NP, PRE>), grid, dim3(64 * WM * WN), 0, st, p);
    else hipLaunchKernelGGL((igemm_conv_x3_kernel<MR, NR, WM, WN, false, false, NP, PRE, true>), grid, dim3(64 * WM * WN), 0, st, p);
}

template <int MR, int NR, int WM, int WN>
void launch_x3_cfg(IgemmParams &p, bool strided, hipStream_t st, int batches = 1) {
    if (p.amax_a && p.amax_b) {
        if (p.b_pairs) launch_x3_cfg_np<MR, NR, WM, WN, 2, true>(p, strided, st, batches);
        else launch_x3_cfg_np<MR, NR, WM, WN, 2>(p, strided, st, batches);
    } else {
        launch_x3_cfg_np<MR, NR, WM, WN, 3>(p, strided, st, batches);
    }
}

// conv_h2p_kernel (igemm_h2p.inc): which launches take it.  Built-in rule: K >= H2P_MIN_K, i.e. the 3x3 layers.  Below (1x1
// convs, K <= 2048) the epilogue (256 KB of output per tile) is a large part of a tile's time and the kernel's single workgroup
// per CU has nothing to overlap it with: measured 0.90-1.0x of the 128x256 kernel there, 1.04-1.17x on the 3x3 layers.
// RCF_CONV_H2P_NEVER / RCF_CONV_H2P_ALWAYS (every eligible shape: tests) override the rule per call.
constexpr int H2P_MIN_K = 2304;
// workgroups per row range: the smallest power of two dividing the column-tile count that gives a workgroup >= 12 row
// blocks (one block of imbalance is then <= 8 %), else the largest one
int h2p_gn(int M, int ntiles) {
    const int RB = rcf_cdiv(M, 32);
    int gn = 1;
    while ((long)RB * gn < 12L * H2P_G && gn * 2 <= 16 && ntiles % (gn * 2) == 0) gn *= 2;
    return gn;
}
int h2p_stat_rows(int M, int gn) {                        // partial statistics rows per row range: sub-tiles of the longest one
    const int GM = H2P_G / gn;
    const int RB = rcf_cdiv(M, 32), base = RB / GM, extra = RB - base * GM;
    int n = h2p_subtiles(extra ? base + 1 : base);
    if (extra && base > 0 && h2p_subtiles(base) > n) n = h2p_subtiles(base);
    return n < 1 ? 1 : n;
}

// is the plane-separated (LDS-image) half of a rcf_conv_weight_pairs2_f32 buffer written for this shape?  Whenever the kernels
// that read it by LDS-DMA (conv_h2p_kernel, conv_h2d_kernel) can take the shape at all: whole K-steps that do not straddle a tap
bool pairs2_written(int K, int Cs) { return K % 16 == 0 && Cs % 16 == 0; }

bool h2p_eligible(const IgemmParams &p, int batches) {
    if ((p.flags & RCF_CONV_H2P_NEVER) || !p.b_pairs2 || !p.amax_a || !p.amax_b || batches != 1 || p.batch1 > 0) return false;
    if (p.div > 1 || p.bias || p.act != 0 || p.Ncol % 256 || p.K % 16 || p.Cs % 16) return false;
    // the statistics workspace is sized for one row of partial sums per 64 GEMM rows (rcf_conv2d_fwd_stats_workspace_bytes)
    const int gn = h2p_gn(p.M, p.Ncol / 256);
    if (p.stats && (long)(H2P_G / gn) * h2p_stat_rows(p.M, gn) > rcf_cdiv(p.M, 64)) return false;
    // built-in rule: the deep 3x3 layers with few column tiles.  Many column tiles on a short K (the data gradient of a
    // 3x3 256 -> 2048 conv: 8 tiles, K = 2304) re-read the activation tile and pay the tile prologue / epilogue once per
    // column tile: measured equal to or 2 % behind the 128 x 256 kernel there
    // ... and many rows (3x3 256 -> 256 at 120x214: 3 210 tiles of the 128 x 256 kernel = 6.3 rounds, little tail left to win,
    // while a workgroup here walks 7 sub-tiles with nothing to overlap their prologues / epilogues: measured 0.92-0.95x)
    return (p.flags & RCF_CONV_H2P_ALWAYS) || (p.K >= H2P_MIN_K && p.M >= 32768 && (long)p.K * 256 >= 1152L * p.Ncol &&
                                               (long)rcf_cdiv(p.M, 128) * (p.Ncol / 256) <= 2048);
}

int launch_h2p(IgemmParams &p, hipStream_t st) {
    const long bytes = (long)(p.K / 16) * p.Ncol * 64;
    const long per_tile_imgs = 256 / (long)p.rr + 2;
    if (bytes >= (1L << 31) || per_tile_imgs * p.a_img_stride * 4 >= (1L << 31)) return RCF_EINVAL;
    p.b_bytes = (int)bytes;
    p.ntiles = p.Ncol / 256;
    p.h2p_gn = h2p_gn(p.M, p.ntiles);
    p.mtiles8 = h2p_stat_rows(p.M, p.h2p_gn);
    p.mtiles = (H2P_G / p.h2p_gn) * p.mtiles8;                          // rows of partial statistics (rcf_conv2d_fwd_bnstats_f32)
    if (p.step < 0) hipLaunchKernelGGL(conv_h2p_kernel<true>, dim3(H2P_G), dim3(256), 0, st, p);
    else hipLaunchKernelGGL(conv_h2p_kernel<false>, dim3(H2P_G), dim3(256), 0, st, p);
    RCF_LAUNCH_CHECK();
    return 0;
}

// conv_h2d_kernel (igemm_h2d.inc): the activation operand arrives as fp16 pair planes -- the only kernel that reads them
int launch_h2d(IgemmParams &p, hipStream_t st, int batches, int *kernel_only) {
    if (!p.b_pairs2 || !p.amax_a || !p.amax_b || batches != 1 || p.batch1 > 0) return RCF_EINVAL;
    if (p.K % 16 || p.Cs % 16 || p.a_pitch != p.Cs || p.Ncol % 4) return RCF_EINVAL;
    const long bytes = (long)(p.K / 16) * p.Ncol * 64;
    const long per_tile_imgs = 256 / (long)p.rr + 2;
    if (bytes >= (1L << 31) || per_tile_imgs * p.a_img_stride * 4 >= (1L << 31)) return RCF_EINVAL;
    if (kernel_only) { *kernel_only = 3; return 0; }
    p.b_bytes = (int)bytes;
    const int BN = p.Ncol > 128 ? 256 : (p.Ncol > 64 ? 128 : 64);
    p.mtiles = rcf_cdiv(p.M, 128);
    p.mtiles8 = rcf_cdiv(p.mtiles, 8);
    p.ntiles = rcf_cdiv(p.Ncol, BN);
    p.colmap = p.Ncol % BN == 0 &&
               rcf_colmap_pays(!(p.flags & RCF_CONV_NO_COLMAP), (long)p.M * p.Cs * 4, (long)p.K * p.Ncol * 4, p.mtiles, p.ntiles);
    const dim3 grid((unsigned)(p.mtiles8 * 8 * p.ntiles));
#define RCF_H2D(NRv)                                                                                            \
    do {                                                                                                        \
        if (p.stats && p.step < 0 && p.div > 1) hipLaunchKernelGGL((conv_h2d_kernel<NRv, true, true, true>), grid, dim3(256), 0, st, p); \
        else if (p.stats && p.step < 0) hipLaunchKernelGGL((conv_h2d_kernel<NRv, false, true, true>), grid, dim3(256), 0, st, p);        \
        else if (p.div > 1) hipLaunchKernelGGL((conv_h2d_kernel<NRv, true, true>), grid, dim3(256), 0, st, p);  \
        else if (p.step < 0) hipLaunchKernelGGL((conv_h2d_kernel<NRv, false, true>), grid, dim3(256), 0, st, p); \
        else hipLaunchKernelGGL((conv_h2d_kernel<NRv, false, false>), grid, dim3(256), 0, st, p);               \
    } while (0)
    if (BN == 256) RCF_H2D(4);
    else if (BN == 128) RCF_H2D(2);
    else RCF_H2D(1);
#undef RCF_H2D
    RCF_LAUNCH_CHECK();
    return 0;
}

// `kernel_only`: do not launch, report which kernel the call would take (rcf_conv_kernel_of: 1 the 128 x 256 family, 2 conv_h2p_kernel,
// 3 conv_h2d_kernel)
int launch_igemm_x3(IgemmParams &p, hipStream_t st, int batches = 1, int *kernel_only = nullptr) {
    p.cs_magic = magic_of(p.Cs);
    {
        const int taps = p.K / p.Cs, kch = rcf_kchunk(korder_chunked(p.flags), taps, p.Cs, RCF_KCHUNK_F32);
        p.kch = kch ? kch : p.Cs;
        p.rsch = taps * p.kch;
        p.kch_magic = magic_of(p.kch);
        p.rsch_magic = magic_of(p.rsch);
        if ((long)(p.K + 16) * p.rsch >= (1L << 32)) return RCF_EINVAL;
    }
    p.s_magic = magic_of(p.S);
    if ((long)p.K * p.Cs >= (1L << 32)) return RCF_EINVAL;
    if (p.a_split) return launch_h2d(p, st, batches, kernel_only);
    if (kernel_only) {
        *kernel_only = h2p_eligible(p, batches) ? 2 : 1;
        return 0;
    }
    if (h2p_eligible(p, batches)) return launch_h2p(p, st);
    // 32-bit descriptor offsets: the images one row tile can touch must lie within 2 GiB of the first one
    const long per_tile_imgs = 256 / (long)p.rr + 2;
    if (per_tile_imgs * p.a_img_stride * 4 >= (1L << 31) || (long)p.Ncol * p.ldb * 4 >= (1L << 31)) return RCF_EINVAL;
    p.b_bytes = (int)((long)p.Ncol * p.ldb * 4);
    if (p.b_pairs && p.amax_a && p.amax_b) {     // K-step-major pairs: K padded to whole steps
        const long bytes = (long)rcf_cdiv(p.K, 16) * p.Ncol * 64;
        if (bytes >= (1L << 31)) return RCF_EINVAL;
        p.b_bytes = (int)bytes;
    }
    const bool strided = p.div > 1;
    if (p.Ncol <= 64 && (long)rcf_cdiv(p.M, 128) * batches < 512) launch_x3_cfg<1, 1, 2, 2>(p, strided, st, batches);   // few rows: 64x64 tiles fill more CUs
    else if (p.Ncol <= 64) launch_x3_cfg<2, 1, 2, 2>(p, strided, st, batches);
    else if (p.Ncol > 128) launch_x3_cfg<2, 4, 2, 2>(p, strided, st, batches);
    else launch_x3_cfg<2, 2, 2, 2>(p, strided, st, batches);
    RCF_LAUNCH_CHECK();
    return 0;
}

template <int BMODE>
int launch_igemm(IgemmParams &p, hipStream_t st) {
    const int v = fp32_mfma_variant(p.flags) < 0 ? 0 : fp32_mfma_variant(p.flags);
    switch (v & 3) {
        case 1: return launch_igemm_v<BMODE, 32, false>(p, st);
        case 2: return launch_igemm_v<BMODE, 16, true>(p, st);
        case 3: return launch_igemm_v<BMODE, 32, true>(p, st);
        default: return launch_igemm_v<BMODE, 16, false>(p, st);
    }
}

inline int region_pixels(const rcf_conv_region *r, int H, int W) {
    if (!r) return H * W;
    return r->band > 0 ? 2 * r->band * r->w + 2 * r->band * (r->h - 2 * r->band) : r->h * r->w;
}

struct WgradPlan {
    int mr, nr, itiles, jtiles, splitk;
    long chunk;
    bool cols;      // igemm_wgrad_h2t_kernel: the taps are GEMM columns (grid.y = 1)
};
WgradPlan plan_wgrad(const rcf_conv_shape *s, const rcf_conv_region *reg = nullptr) {
    WgradPlan pl;
    const bool x3 = use_x3(s->flags);
    const bool smallc = s->Cin == 4;
    const int ncols = smallc ? s->R * s->S * 4 : s->Cin;
    pl.mr = s->Cout > 64 ? 2 : 1;
    pl.nr = ncols > 64 ? 2 : 1;
    // 128 x 256 tile (igemm_wgrad_x3_wide_kernel): whole tensors, wide enough operands
    const bool wide = x3 && !smallc && !reg && s->Cout >= 128 && s->Cin >= 256;
    if (wide) pl.nr = 4;
    // fp16 pairs with the (tap, channel) pairs as GEMM columns (igemm_wgrad_h2t_kernel): 128 x 256 tiles over R*S*Cin
    // columns (128 x 128 below 256 columns), whole tensors and regions, from 64 output channels and 64 columns up: the
    // narrow layers are bound by memory, not by the half-empty tiles (1x1 64->256 and 256->64 at 120x214: 0.19 -> 0.12 ms
    // against the 64-wide kernels).
    const int ktot = s->R * s->S * s->Cin;
    pl.cols = s->amax_dy && s->amax_x && x3 && !smallc && s->Cin % 64 == 0 && s->Cout >= 64 && !(reg && ktot < 256);
    if (pl.cols) {
        pl.mr = 2;
        pl.nr = ktot >= 256 ? 4 : 2;
        // 256 x 256 tile, one workgroup per CU: whole 256-row and 256-column tiles of one tap only
        if (!(s->flags & (RCF_CONV_WGRAD_TILE_128 | RCF_CONV_X_PLANES | RCF_CONV_DY_PLANES)) && pl.nr == 4 && s->Cout % 256 == 0 && s->Cin % 256 == 0) pl.mr = 4;
    }
    pl.itiles = rcf_cdiv(s->Cout, 64 * pl.mr);
    pl.jtiles = pl.cols ? rcf_cdiv(ktot, 64 * pl.nr) : rcf_cdiv(ncols, 64 * pl.nr);
    const long RR = region_pixels(reg, s->Ho, s->Wo);                       // contributing pixels per image
    const long M = (long)s->N * RR;
    const long tiles = (long)pl.itiles * pl.jtiles * ((smallc || pl.cols) ? 1 : s->R * s->S);
    long sk = (1536 + tiles - 1) / tiles;
    const long maxsk = M / 1024 > 1 ? M / 1024 : 1;
    if (sk > maxsk) sk = maxsk;
    if (sk > 256) sk = 256;
    if (sk < 1) sk = 1;
    if (x3 && !smallc && pl.mr >= 2 && pl.nr >= 2) {
        // the split-bf16 kernel runs 3 workgroups per CU (768 slots): pick the split whose last round is fullest
        // (time ~ rounds / split; the fixed-order reduction costs ~ split)
        const long slots = pl.mr == 4 ? 256 : (pl.nr == 4 ? 512 : 768), hi = maxsk < 256 ? maxsk : 256;   // (the 128 x 256 kernel: 2 per CU = 512; 256 x 256: 1)
        double best = 1e30;
        {
            // the cost model of the bf16 weight gradient (csrc/igemm_bf16.hip), in microseconds: rounds x pixels per
            // workgroup x time per pixel (three partial products: 3 x the bf16 figure) + the fixed-order reduction, which
            // reads c copies of the weight gradient
            const double px_us = 0.075 * fmax((double)(pl.mr * pl.nr) / 8.0, 0.35);
            const double wbytes = (double)s->Cout * s->R * s->S * s->Cin * 4.0;
            for (long c = 1; c <= hi; ++c) {
                const double rounds = (double)((tiles * c + slots - 1) / slots);
                const double cost = rounds * (double)((M + c - 1) / c) * px_us + (c > 1 ? (double)c * wbytes / 2.0e6 + 3.0 : 0.0);
                if (cost < best - 1e-9) { best = cost; sk = c; }
            }
        }
    }
    long chunk = (M + sk - 1) / sk;
    chunk = (chunk + BK - 1) / BK * BK;
    if (x3 && !smallc) {
        // 32-bit descriptor offsets: the images (and dy rows) one pixel chunk touches must span < 2 GiB
        const long img_bytes = (long)s->H * s->W * s->x_pitch * 4, dy_bytes = (long)s->Ho * s->Wo * s->y_pitch * 4;
        while (chunk > BK && ((chunk / RR + 2) * img_bytes >= (1L << 31) || (chunk / RR + 2) * dy_bytes >= (1L << 31)))
            chunk = (chunk / 2 + BK - 1) / BK * BK;
    }
    sk = (M + chunk - 1) / chunk;
    pl.splitk = (int)sk;
    pl.chunk = chunk;
    return pl;
}

}  // namespace

namespace {
// rectangle (or frame) of the GEMM-row tensor [N, H, W]; null = everything.  Returns 0 / RCF_EINVAL.
int set_region(IgemmParams &p, const rcf_conv_region *r, int N, int H, int W) {
    p.ry0 = r ? r->y0 : 0; p.rx0 = r ? r->x0 : 0; p.rh = r ? r->h : H; p.rw = r ? r->w : W;
    p.rband = r ? r->band : 0;
    if (p.ry0 < 0 || p.rx0 < 0 || p.rh <= 0 || p.rw <= 0 || p.ry0 + p.rh > H || p.rx0 + p.rw > W) return RCF_EINVAL;
    if (p.rband < 0 || (p.rband > 0 && (2 * p.rband >= p.rh || 2 * p.rband >= p.rw))) return RCF_EINVAL;
    p.rr = region_pixels(r, H, W);
    p.M = N * p.rr;
    return 0;
}
}  // namespace

/* C[M][N] (pitch ldc) (+)= A[M][K] (pitch lda) . B[N][K]^T (pitch ldb) + bias[N], then act (0 none, 1 LeakyReLU,
 * 2 GELU): nn.Linear / attention products of the DINO ViT (models/dino_vit.py:110-134) on the split-bf16 conv
 * kernel -- a 1x1 convolution whose "pixels" are the M rows. */
extern "C" int rcf_gemm_nt_f32(const float *A, int lda, const float *B, int ldb, const float *bias, float *C, int ldc,
                               int M, int N, int K, int act, float slope, int beta, const unsigned *amax_a,
                               const unsigned *amax_b, const void *b_pairs, unsigned *amax_out, void *stream) {
    if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0 || K % 4 || lda % 4 || ldb % 4 || ldc % 4) return RCF_EINVAL;
    if (lda < K || ldb < K || ldc < N || !rcf_aligned16(A) || !rcf_aligned16(B) || !rcf_aligned16(C)) return RCF_EINVAL;
    if ((long)M * lda >= (1L << 29) || (long)N * ldb >= (1L << 29)) return RCF_EINVAL;      // 32-bit descriptor offsets
    IgemmParams p{};
    p.A = A; p.Bw = B; p.bias = bias; p.Y = C;
    p.M = M; p.Ncol = N; p.K = K;
    p.Ho = M; p.Wo = 1; p.Hs = M; p.Ws = 1; p.Cs = K; p.S = 1;
    p.up = 1; p.off = 0; p.step = 1; p.div = 1;
    p.a_pitch = lda; p.a_img_stride = (long)M * lda; p.y_pitch = ldc;
    p.ldb = ldb; p.act = act; p.slope = slope; p.beta = beta;
    p.ry0 = 0; p.rx0 = 0; p.rh = M; p.rw = 1; p.rband = 0; p.rr = M;
    // operand ranges -> fp16-pair kernels; B already split (rcf_conv_weight_pairs_f32 with Cout = N, Cin = K, R = S = 1)
    // needs the plain layout it was made from (ldb == K)
    if (b_pairs && ldb != K) return RCF_EINVAL;
    p.amax_a = amax_a; p.amax_b = amax_b; p.b_pairs = (amax_a && amax_b) ? b_pairs : nullptr; p.amax_out = amax_out;
    return launch_igemm_x3(p, rcf_stream(stream));
}

/* batch0 x batch1 independent products in one launch (attention: images x heads): operand / result of product (i0, i1)
 * start at A + i0*a_s0 + i1*a_s1 etc. (element strides) */
extern "C" int rcf_gemm_nt_batched_f32(const float *A, int lda, long a_s0, long a_s1, const float *B, int ldb, long b_s0,
                                       long b_s1, float *C, int ldc, long c_s0, long c_s1, int batch0, int batch1, int M,
                                       int N, int K, int act, float slope, int beta, void *stream) {
    if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0 || K % 4 || lda % 4 || ldb % 4 || ldc % 4) return RCF_EINVAL;
    if (lda < K || ldb < K || ldc < N || !rcf_aligned16(A) || !rcf_aligned16(B) || !rcf_aligned16(C)) return RCF_EINVAL;
    if (batch0 <= 0 || batch1 <= 0 || (long)batch0 * batch1 > 65535) return RCF_EINVAL;
    if ((a_s0 | a_s1 | b_s0 | b_s1 | c_s0 | c_s1) % 4) return RCF_EINVAL;
    if ((long)M * lda >= (1L << 29) || (long)N * ldb >= (1L << 29)) return RCF_EINVAL;
    IgemmParams p{};
    p.A = A; p.Bw = B; p.bias = nullptr; p.Y = C;
    p.M = M; p.Ncol = N; p.K = K;
    p.Ho = M; p.Wo = 1; p.Hs = M; p.Ws = 1; p.Cs = K; p.S = 1;
    p.up = 1; p.off = 0; p.step = 1; p.div = 1;
    p.a_pitch = lda; p.a_img_stride = (long)M * lda; p.y_pitch = ldc;
    p.ldb = ldb; p.act = act; p.slope = slope; p.beta = beta;
    p.ry0 = 0; p.rx0 = 0; p.rh = M; p.rw = 1; p.rband = 0; p.rr = M;
    p.batch1 = batch1; p.a_bs0 = a_s0; p.a_bs1 = a_s1; p.b_bs0 = b_s0; p.b_bs1 = b_s1; p.y_bs0 = c_s0; p.y_bs1 = c_s1;
    return launch_igemm_x3(p, rcf_stream(stream), batch0 * batch1);
}

extern "C" int rcf_conv2d_fwd_f32(const float *x, const float *w, const float *bias, float *y,
                                  const rcf_conv_shape *s, int act, float slope, int beta, void *stream) {
    return rcf_conv2d_fwd_region_f32(x, w, bias, y, s, nullptr, act, slope, beta, stream);
}

namespace {
int conv2d_dgrad_impl(const float *dy, const float *w, float *dx, const rcf_conv_shape *s, const rcf_conv_region *region, int beta,
                      void *workspace, size_t workspace_bytes, void *stream, int *kernel_only, const rcf_bn_bwd_in *bn = nullptr,
                      double *stats = nullptr, int *mtiles_out = nullptr, const float *add = nullptr, int add_pitch = 0,
                      const unsigned char *add_mask = nullptr);
// forward launch; kernel_only: report the kernel the call would take instead (rcf_conv_kernel_of)
int conv2d_fwd_impl(const float *x, const float *w, const float *bias, float *y, const rcf_conv_shape *s,
                    const rcf_conv_region *region, int act, float slope, int beta, double *stats, void *stream, int *kernel_only,
                    int *mtiles_out = nullptr) {
    IgemmParams p{};
    p.flags = s->flags;
    p.A = x; p.Bw = w; p.bias = bias; p.Y = y;
    p.Ncol = s->Cout; p.K = s->R * s->S * s->Cin;
    p.Ho = s->Ho; p.Wo = s->Wo; p.Hs = s->H; p.Ws = s->W; p.Cs = s->Cin; p.S = s->S;
    if (int e = set_region(p, region, s->N, s->Ho, s->Wo)) return e;
    p.up = s->stride; p.off = -s->pad; p.step = s->dil; p.div = 1;
    p.a_pitch = s->x_pitch; p.a_img_stride = (long)s->H * s->W * s->x_pitch; p.y_pitch = s->y_pitch;
    p.ldb = p.K; p.act = act; p.slope = slope; p.beta = beta;
    p.amax_a = s->amax_x; p.amax_b = s->amax_w; p.b_pairs = s->w_pairs;
    if (s->w_pairs2) {
        p.b_pairs = s->w_pairs2;
        if (pairs2_written(p.K, s->Cin))
            p.b_pairs2 = (const char *)s->w_pairs2 + rcf_conv_weight_pairs_bytes(s->Cout, s->Cin, s->R, s->S);
    }
    p.stats = stats;
    p.amax_out = s->amax_y;
    p.a_split = (s->flags & RCF_CONV_X_PLANES) ? 1 : 0;
    if (use_x3(s->flags)) {
        const int e = launch_igemm_x3(p, rcf_stream(stream), 1, kernel_only);
        if (mtiles_out) *mtiles_out = p.mtiles;
        return e;
    }
    if (region || stats) return RCF_EINVAL;              // sub-rectangles / fused statistics exist on the default kernels only
    if (kernel_only) { *kernel_only = 0; return 0; }
    return launch_igemm<0>(p, rcf_stream(stream));
}
}  // namespace

extern "C" int rcf_conv2d_fwd_region_f32(const float *x, const float *w, const float *bias, float *y,
                                         const rcf_conv_shape *s, const rcf_conv_region *region, int act, float slope,
                                         int beta, void *stream) {
    if (int e = check_shape(s)) return e;
    if (!x || !w || !y || !rcf_aligned16(x) || !rcf_aligned16(w) || !rcf_aligned16(y)) return RCF_EINVAL;
    if (!region && act == 0 && thin_path(s)) return rcf_thin_fwd(x, w, bias, y, s, beta, rcf_stream(stream));
    return conv2d_fwd_impl(x, w, bias, y, s, region, act, slope, beta, nullptr, stream, nullptr);
}

/* which kernel family a forward (dgrad = 0) or data-gradient (dgrad = 1) launch with this shape, these operand pointers and
 * flags takes -- 0 the fp32-MFMA kernels, 1 the 128 x 256-tile family, 2 conv_h2p_kernel (persistent LDS-DMA) -- without launching
 * anything: a pure function of its arguments (profiling labels; it replaces a "last kernel" global). */
extern "C" int rcf_conv_kernel_of(const rcf_conv_shape *s, const rcf_conv_region *region, int dgrad) {
    if (check_shape(s)) return RCF_EINVAL;
    int k = RCF_EINVAL;
    float dummy = 0.f;
    const int e = dgrad ? conv2d_dgrad_impl(&dummy, &dummy, &dummy, s, region, 0, &dummy, (size_t)1 << 40, nullptr, &k)
                        : conv2d_fwd_impl(&dummy, &dummy, nullptr, &dummy, s, region, 0, 0.f, 0, nullptr, nullptr, &k);
    return e ? e : k;
}

extern "C" int rcf_absmax_f32(const float *x, long rows, int C, int pitch, unsigned *amax, void *stream) {
    if (!x || !amax || rows <= 0 || C <= 0 || C % 4 || pitch % 4 || pitch < C || !rcf_aligned16(x)) return RCF_EINVAL;
    const long items = rows * (C / 4);
    const long blocks = (items + 1023) / 1024;          // >= 4 float4 per thread
    hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), 0, rcf_stream(stream), x,
                       rows, C, pitch, amax);
    RCF_LAUNCH_CHECK();
    return 0;
}

extern "C" size_t rcf_conv_weight_pairs_bytes(int Cout, int Cin, int R, int S) {
    if (Cout <= 0 || Cin <= 0 || R <= 0 || S <= 0) return 0;
    return (size_t)rcf_cdiv(R * S * Cin, 16) * Cout * 64;
}

extern "C" int rcf_conv_weight_pairs_f32(const float *w, int Cout, int Cin, int R, int S, const unsigned *amax_w,
                                         void *planes, unsigned flags, void *stream) {
    if (!w || !amax_w || !planes || Cout <= 0 || Cin <= 0 || R <= 0 || S <= 0 || !rcf_aligned16(planes)) return RCF_EINVAL;
    const long n = (long)Cout * R * S * Cin;
    const long blocks = (n + 1023) / 1024;
    hipLaunchKernelGGL(weight_pairs_kernel<false>, dim3((unsigned)(blocks < 1024 ? blocks : 1024)), dim3(256), 0,
                       rcf_stream(stream), w, amax_w, (_Float16 *)planes, Cout, Cin, R * S, (int)korder_chunked(flags));
    RCF_LAUNCH_CHECK();
    return 0;
}

extern "C" int rcf_conv_weight_pairs_t_f32(const float *w, int Cout, int Cin, int R, int S, const unsigned *amax_w,
                                           void *planes, unsigned flags, void *stream) {
    if (!w || !amax_w || !planes || Cout <= 0 || Cin <= 0 || R <= 0 || S <= 0 || !rcf_aligned16(planes)) return RCF_EINVAL;
    const dim3 tgrid(rcf_cdiv(Cin, 32), rcf_cdiv(Cout, 32), R * S);
    hipLaunchKernelGGL(weight_pairs_kernel<true>, tgrid, dim3(256), 0, rcf_stream(stream), w, amax_w, (_Float16 *)planes,
                       Cout, Cin, R * S, (int)korder_chunked(flags));
    RCF_LAUNCH_CHECK();
    return 0;
}

/* Both layouts of the split weights in one buffer -- [pairs (rcf_conv_weight_pairs_f32 / _t_f32) | pairs2 (plane-separated
 * K-step blocks for the LDS-DMA loads of conv_h2p_kernel)] -- for rcf_conv_shape.w_pairs2 (transpose = 0: forward) and
 * .w_pairs2_t (transpose = 1: data gradient). */
extern "C" size_t rcf_conv_weight_pairs2_bytes(int Cout, int Cin, int R, int S, int transpose) {
    if (Cout <= 0 || Cin <= 0 || R <= 0 || S <= 0) return 0;
    const size_t one = transpose ? (size_t)rcf_cdiv(R * S * Cout, 16) * 16 * Cin * sizeof(float)
                                 : rcf_conv_weight_pairs_bytes(Cout, Cin, R, S);
    return 2 * one;
}

extern "C" int rcf_conv_weight_pairs2_f32(const float *w, int Cout, int Cin, int R, int S, int transpose,
                                          const unsigned *amax_w, void *planes, unsigned flags, void *stream) {
    if (!w || !amax_w || !planes || Cout <= 0 || Cin <= 0 || R <= 0 || S <= 0 || !rcf_aligned16(planes)) return RCF_EINVAL;
    const size_t one = rcf_conv_weight_pairs2_bytes(Cout, Cin, R, S, transpose) / 2;
    char *second = (char *)planes + one;
    const int K = R * S * (transpose ? Cout : Cin);
    // the second half is written for every shape a kernel reading it can take (a pure function of the shape: whole K-steps)
    const bool second_half = pairs2_written(K, transpose ? Cout : Cin);
    const int korder = (int)korder_chunked(flags);
    if (transpose) {
        if (int e = rcf_conv_weight_pairs_t_f32(w, Cout, Cin, R, S, amax_w, planes, flags, stream)) return e;
        const dim3 tgrid(rcf_cdiv(Cin, 32), rcf_cdiv(Cout, 32), R * S);
        if (second_half)
            hipLaunchKernelGGL(weight_pairs2_kernel<true>, tgrid, dim3(256), 0, rcf_stream(stream), w, amax_w, (_Float16 *)second,
                               Cout, Cin, R * S, korder);
    } else {
        if (int e = rcf_conv_weight_pairs_f32(w, Cout, Cin, R, S, amax_w, planes, flags, stream)) return e;
        const long n = (long)Cout * R * S * Cin;
        const long blocks = (n + 1023) / 1024;
        if (second_half)
            hipLaunchKernelGGL(weight_pairs2_kernel<false>, dim3((unsigned)(blocks < 1024 ? blocks : 1024)), dim3(256), 0,
                               rcf_stream(stream), w, amax_w, (_Float16 *)second, Cout, Cin, R * S, korder);
    }
    RCF_LAUNCH_CHECK();
    return 0;
}

/* does rcf_conv_weight_pairs2_f32 write the plane-separated half for this shape (and may a launch read it)?  A pure function
 * of the shape; for callers that fill rcf_wprep_entry.flags */
extern "C" int rcf_conv_pairs2_useful(int Cout, int Cin, int R, int S, int transpose) {
    return pairs2_written(R * S * (transpose ? Cout : Cin), transpose ? Cout : Cin) ? 1 : 0;
}

/* Batched weight preparation of the fp16-pair kernels: ranges (amax), then both rcf_conv_weight_pairs2_f32 buffers of every
 * weight, in THREE launches for the whole model.  tab_*: device arrays of n rcf_wprep_entry (csrc/rcf_common.h) with
 * first_block / nblocks filled per launch (blocks_* = their totals); amax_base: the n consecutive range slots the entries
 * point to (zeroed here).  Entry i of every table describes the same weight.  Identical bytes to the per-weight calls. */
extern "C" int rcf_conv_weights_prepare_f32(const void *tab_absmax, int blocks_absmax, const void *tab_pairs, int blocks_pairs,
                                            const void *tab_pairs_t, int blocks_pairs_t, int n, unsigned *amax_base, unsigned flags,
                                            void *stream) {
    if (!tab_absmax || !tab_pairs || !tab_pairs_t || n <= 0 || !amax_base || blocks_absmax <= 0 || blocks_pairs <= 0 || blocks_pairs_t <= 0)
        return RCF_EINVAL;
    hipStream_t st = rcf_stream(stream);
    if (hipMemsetAsync(amax_base, 0, (size_t)n * sizeof(unsigned), st) != hipSuccess) return RCF_EINVAL;
    hipLaunchKernelGGL(wprep_absmax_kernel, dim3((unsigned)blocks_absmax), dim3(256), 0, st, (const rcf_wprep_entry *)tab_absmax, n);
    hipLaunchKernelGGL(wprep_pairs_kernel, dim3((unsigned)blocks_pairs), dim3(256), 0, st, (const rcf_wprep_entry *)tab_pairs, n, (int)korder_chunked(flags));
    hipLaunchKernelGGL(wprep_pairs_t_kernel, dim3((unsigned)blocks_pairs_t), dim3(256), 0, st, (const rcf_wprep_entry *)tab_pairs_t, n, (int)korder_chunked(flags));
    RCF_LAUNCH_CHECK();
    return 0;
}

extern "C" size_t rcf_conv2d_fwd_stats_workspace_bytes(const rcf_conv_shape *s) {
    if (check_shape(s) || !use_x3(s->flags)) return 0;
    // one row of partial sums per row tile (smallest tile: 64 rows) + the 64 rows of the two-level reduction
    return (size_t)(rcf_cdiv((long)s->N * s->Ho * s->Wo, 64) + 64) * 2 * s->Cout * sizeof(double);
}

extern "C" int rcf_conv2d_fwd_bnstats_f32(const float *x, const float *w, float *y, const rcf_conv_shape *s, double *sums,
                                          const rcf_bn_finalize *fin, void *workspace, size_t workspace_bytes,
                                          void *stream) {
    if (int e = check_shape(s)) return e;
    if (!x || !w || !y || (!sums && !fin) || !rcf_aligned16(x) || !rcf_aligned16(w) || !rcf_aligned16(y)) return RCF_EINVAL;
    if (!use_x3(s->flags)) return RCF_EINVAL;             // the statistics epilogue exists on the default kernels only
    if (!workspace || workspace_bytes < rcf_conv2d_fwd_stats_workspace_bytes(s)) return RCF_EWORKSPACE;
    int mtiles = 0;
    if (int e = conv2d_fwd_impl(x, w, nullptr, y, s, nullptr, 0, 0.f, 0, (double *)workspace, stream, nullptr, &mtiles)) return e;
    return rcf_sum_partials_bn((const double *)workspace, mtiles, s->Cout, sums,
                               (double *)workspace + (size_t)mtiles * 2 * s->Cout, fin, stream);
}

extern "C" int rcf_conv2d_fwd_stats_f32(const float *x, const float *w, float *y, const rcf_conv_shape *s, double *sums,
                                        void *workspace, size_t workspace_bytes, void *stream) {
    return rcf_conv2d_fwd_bnstats_f32(x, w, y, s, sums, nullptr, workspace, workspace_bytes, stream);
}

extern "C" size_t rcf_conv2d_dgrad_workspace_bytes(const rcf_conv_shape *s) {
    if (check_shape(s) || !use_x3(s->flags)) return 0;
    // transposed fp32 weights, or their fp16 pairs with K = R*S*Cout padded to whole K-steps
    return (size_t)rcf_cdiv(s->R * s->S * s->Cout, 16) * 16 * s->Cin * sizeof(float);
}

extern "C" int rcf_conv2d_dgrad_f32(const float *dy, const float *w, float *dx, const rcf_conv_shape *s, int beta,
                                    void *workspace, size_t workspace_bytes, void *stream) {
    return rcf_conv2d_dgrad_region_f32(dy, w, dx, s, nullptr, beta, workspace, workspace_bytes, stream);
}

extern "C" int rcf_conv2d_dgrad_region_f32(const float *dy, const float *w, float *dx, const rcf_conv_shape *s,
                                           const rcf_conv_region *region, int beta, void *workspace,
                                           size_t workspace_bytes, void *stream) {
    if (int e = check_shape(s)) return e;
    if (!dy || !w || !dx || !rcf_aligned16(dy) || !rcf_aligned16(w) || !rcf_aligned16(dx)) return RCF_EINVAL;
    if (!region && thin_path(s)) return rcf_thin_dgrad(dy, w, dx, s, beta, rcf_stream(stream));
    return conv2d_dgrad_impl(dy, w, dx, s, region, beta, workspace, workspace_bytes, stream, nullptr);
}

extern "C" size_t rcf_conv2d_dgrad_bnsums_workspace_bytes(const rcf_conv_shape *s) {
    if (check_shape(s) || !use_x3(s->flags)) return 0;
    // one row of partial sums per row tile (smallest tile: 64 rows) + the 64 rows of the two-level reduction
    return (size_t)(rcf_cdiv((long)s->N * s->H * s->W, 64) + 64) * 2 * s->Cin * sizeof(double);
}

extern "C" int rcf_conv2d_dgrad_bnsums_ok(const rcf_conv_shape *s) {
    if (check_shape(s) || !use_x3(s->flags) || s->Cout % 4) return 0;
    const int bnw = s->Cin > 128 ? 256 : (s->Cin > 64 ? 128 : 64);
    const void *wpt = s->w_pairs2_t ? s->w_pairs2_t : s->w_pairs_t;
    if (s->Cin % bnw || !wpt || !s->amax_dy || !s->amax_w) return 0;
    if ((s->flags & RCF_CONV_DY_PLANES) && !(s->w_pairs2_t && pairs2_written(s->R * s->S * s->Cout, s->Cout))) return 0;
    return 1;
}

extern "C" int rcf_conv2d_dgrad_bnsums_f32(const float *dy, const float *w, float *dx, const rcf_conv_shape *s, int beta,
                                           const rcf_bn_bwd_in *bn, double *sums2, void *workspace, size_t workspace_bytes,
                                           void *stream) {
    return rcf_conv2d_dgrad_add_f32(dy, w, dx, s, beta, nullptr, 0, nullptr, bn, sums2, workspace, workspace_bytes, stream);
}

extern "C" int rcf_conv2d_dgrad_add_f32(const float *dy, const float *w, float *dx, const rcf_conv_shape *s, int beta, const float *add,
                                        int add_pitch, const unsigned char *add_mask, const rcf_bn_bwd_in *bn, double *sums2,
                                        void *workspace, size_t workspace_bytes, void *stream) {
    if (int e = check_shape(s)) return e;
    if (!dy || !w || !dx || (!bn && !add) || (bn && !sums2) || (add && beta) || !rcf_aligned16(dy) || !rcf_aligned16(w) || !rcf_aligned16(dx))
        return RCF_EINVAL;
    if (!rcf_conv2d_dgrad_bnsums_ok(s)) return RCF_EINVAL;
    if (bn && (!workspace || workspace_bytes < rcf_conv2d_dgrad_bnsums_workspace_bytes(s) || !rcf_aligned16(workspace))) return RCF_EWORKSPACE;
    int mtiles = 0;
    if (int e = conv2d_dgrad_impl(dy, w, dx, s, nullptr, beta, nullptr, 0, stream, nullptr, bn, bn ? (double *)workspace : nullptr, &mtiles,
                                  add, add_pitch, add_mask))
        return e;
    if (!bn) return 0;
    return rcf_sum_partials_f64((const double *)workspace, mtiles, 2 * s->Cin, sums2, (double *)workspace + (size_t)mtiles * 2 * s->Cin,
                                stream);
}

namespace {
int conv2d_dgrad_impl(const float *dy, const float *w, float *dx, const rcf_conv_shape *s, const rcf_conv_region *region, int beta,
                      void *workspace, size_t workspace_bytes, void *stream, int *kernel_only, const rcf_bn_bwd_in *bn, double *stats,
                      int *mtiles_out, const float *add, int add_pitch, const unsigned char *add_mask) {
    if (s->Cout % 4) return RCF_EINVAL;
    IgemmParams p{};
    p.flags = s->flags;
    if (add) {
        // the masked addend lives in the lean epilogue of the fp16-pair kernels, like the batch-norm sums below: same conditions
        const int bnw = s->Cin > 128 ? 256 : (s->Cin > 64 ? 128 : 64);
        const void *wpt = s->w_pairs2_t ? s->w_pairs2_t : s->w_pairs_t;
        if (!use_x3(s->flags) || region || beta || s->Cin % bnw || !wpt || !s->amax_dy || !s->amax_w || !add_mask || add_pitch % 4 ||
            add_pitch < s->Cin || !rcf_aligned16(add))
            return RCF_EINVAL;
        p.flags |= RCF_CONV_H2P_NEVER;
        p.add_src = add; p.add_pitch = add_pitch; p.add_mask = add_mask;
    }
    if (bn) {
        // the batch-norm sums come out of the lean epilogue of the 128-row fp16-pair kernels: whole column tiles, the whole tensor,
        // weights prepared by the caller (the workspace holds the partial sums), not the persistent kernel (its own epilogue)
        const int bnw = s->Cin > 128 ? 256 : (s->Cin > 64 ? 128 : 64);
        const void *wpt = s->w_pairs2_t ? s->w_pairs2_t : s->w_pairs_t;
        if (!use_x3(s->flags) || region || s->Cin % bnw || !wpt || !s->amax_dy || !s->amax_w || !stats || !bn->x || !bn->relu_mask ||
            !bn->mean || !bn->invstd || bn->x_pitch % 4 || bn->x_pitch < s->Cin || !rcf_aligned16(bn->x) || !rcf_aligned16(bn->mean) ||
            !rcf_aligned16(bn->invstd))
            return RCF_EINVAL;
        p.flags |= RCF_CONV_H2P_NEVER;
        p.stats = stats;
        p.bn_x = bn->x; p.bn_x_pitch = bn->x_pitch; p.bn_mask = bn->relu_mask; p.bn_mean = bn->mean; p.bn_invstd = bn->invstd;
    }
    p.A = dy; p.Bw = w; p.bias = nullptr; p.Y = dx;
    p.Ncol = s->Cin; p.K = s->R * s->S * s->Cout;
    p.Ho = s->H; p.Wo = s->W; p.Hs = s->Ho; p.Ws = s->Wo; p.Cs = s->Cout; p.S = s->S;
    if (int e = set_region(p, region, s->N, s->H, s->W)) return e;
    p.up = 1; p.off = s->pad; p.step = -s->dil; p.div = s->stride;
    p.a_pitch = s->y_pitch; p.a_img_stride = (long)s->Ho * s->Wo * s->y_pitch; p.y_pitch = s->x_pitch;
    p.ldb = s->R * s->S * s->Cin; p.act = 0; p.slope = 0.f; p.beta = beta;
    if (use_x3(s->flags)) {
        // k-contiguous weights for the bf16 operand fetch: wt[c][rs][co] (one small transpose per call)
        const size_t need = rcf_conv2d_dgrad_workspace_bytes(s);
        const void *wpt = s->w_pairs2_t ? s->w_pairs2_t : s->w_pairs_t;
        const bool prepared = wpt && s->amax_dy && s->amax_w;
        p.a_split = (s->flags & RCF_CONV_DY_PLANES) ? 1 : 0;
        p.amax_out = s->amax_y;
        if (!prepared && (!workspace || workspace_bytes < need || !rcf_aligned16(workspace))) return RCF_EWORKSPACE;
        hipStream_t st = rcf_stream(stream);
        p.ldb = p.K;
        p.amax_a = s->amax_dy; p.amax_b = s->amax_w;
        const dim3 tgrid(rcf_cdiv(s->Cin, 32), rcf_cdiv(s->Cout, 32), s->R * s->S);
        if (p.amax_a && p.amax_b && wpt) {
            p.b_pairs = wpt;                               // prepared once per weight update by the caller
            if (s->w_pairs2_t && pairs2_written(p.K, s->Cout)) p.b_pairs2 = (const char *)wpt + need;
        } else if (p.amax_a && p.amax_b) {                // fp16 pairs: transposed AND split, once per launch
            if (!kernel_only)
                hipLaunchKernelGGL(weight_pairs_kernel<true>, tgrid, dim3(256), 0, st, w, s->amax_w, (_Float16 *)workspace,
                                   s->Cout, s->Cin, s->R * s->S, (int)korder_chunked(s->flags));
            p.b_pairs = workspace;
        } else {
            if (!kernel_only)
                hipLaunchKernelGGL(weight_transpose_kernel, tgrid, dim3(256), 0, st, w, (float *)workspace, s->Cout, s->Cin,
                                   s->R * s->S);
            p.Bw = (const float *)workspace;
        }
        const int e = launch_igemm_x3(p, st, 1, kernel_only);
        if (mtiles_out) *mtiles_out = p.mtiles;
        return e;
    }
    if (region || bn) return RCF_EINVAL;
    if (kernel_only) { *kernel_only = 0; return 0; }
    return launch_igemm<1>(p, rcf_stream(stream));
}
}  // namespace

namespace {
bool region_ok(const rcf_conv_region *r, int H, int W) {
    return !r || (r->y0 >= 0 && r->x0 >= 0 && r->h > 0 && r->w > 0 && r->y0 + r->h <= H && r->x0 + r->w <= W &&
                  r->band >= 0 && (r->band == 0 || (2 * r->band < r->h && 2 * r->band < r->w)));
}
}  // namespace

extern "C" size_t rcf_conv2d_wgrad_workspace_bytes(const rcf_conv_shape *s) {
    return rcf_conv2d_wgrad_region_workspace_bytes(s, nullptr);
}

extern "C" size_t rcf_conv2d_wgrad_region_workspace_bytes(const rcf_conv_shape *s, const rcf_conv_region *region) {
    if (check_shape(s) || !region_ok(region, s->Ho, s->Wo)) return 0;
    if (!region && thin_path(s)) return rcf_thin_wgrad_workspace_bytes(s);
    const WgradPlan pl = plan_wgrad(s, region);
    if (pl.splitk <= 1) return 0;
    return (size_t)pl.splitk * s->Cout * s->R * s->S * s->Cin * sizeof(float);
}

extern "C" int rcf_conv2d_wgrad_f32(const float *x, const float *dy, float *dw, const rcf_conv_shape *s, int beta,
                                    void *workspace, size_t workspace_bytes, void *stream) {
    return rcf_conv2d_wgrad_region_f32(x, dy, dw, s, nullptr, beta, workspace, workspace_bytes, stream);
}

extern "C" int rcf_conv2d_wgrad_region_f32(const float *x, const float *dy, float *dw, const rcf_conv_shape *s,
                                           const rcf_conv_region *region, int beta, void *workspace,
                                           size_t workspace_bytes, void *stream) {
    if (int e = check_shape(s)) return e;
    if (!x || !dy || !dw || !rcf_aligned16(x) || !rcf_aligned16(dy) || !rcf_aligned16(dw)) return RCF_EINVAL;
    if (s->Cout % 4 || !region_ok(region, s->Ho, s->Wo)) return RCF_EINVAL;
    if (!region && thin_path(s)) return rcf_thin_wgrad(x, dy, dw, s, beta, workspace, workspace_bytes, rcf_stream(stream));
    const WgradPlan pl = plan_wgrad(s, region);
    const size_t need = rcf_conv2d_wgrad_region_workspace_bytes(s, region);
    if (need > 0 && (!workspace || workspace_bytes < need || !rcf_aligned16(workspace))) return RCF_EWORKSPACE;
    hipStream_t st = rcf_stream(stream);
    WgradParams p{};
    p.X = x; p.DY = dy;
    p.OUT = pl.splitk > 1 ? (float *)workspace : dw;
    p.Cout = s->Cout; p.Cin = s->Cin; p.R = s->R; p.S = s->S;
    p.H = s->H; p.W = s->W; p.Ho = s->Ho; p.Wo = s->Wo; p.stride = s->stride; p.pad = s->pad; p.dil = s->dil;
    p.x_pitch = s->x_pitch; p.dy_pitch = s->y_pitch;
    p.ry0 = region ? region->y0 : 0; p.rx0 = region ? region->x0 : 0;
    p.rh = region ? region->h : s->Ho; p.rw = region ? region->w : s->Wo;
    p.rband = region ? region->band : 0; p.rr = region_pixels(region, s->Ho, s->Wo);
    p.M = (long)s->N * p.rr; p.chunk = pl.chunk; p.itiles = pl.itiles; p.jtiles = pl.jtiles;
    p.split_stride = (long)s->Cout * s->R * s->S * s->Cin; p.beta = beta;
    p.amax_a = s->amax_dy; p.amax_b = s->amax_x;
    p.xcd_map = (s->flags & RCF_CONV_NO_WGRAD_XCD) ? 0 : 1;
    const bool smallc = s->Cin == 4;
    const bool x3 = use_x3(s->flags);
    if (region && (smallc || !x3)) return RCF_EINVAL;    // sub-rectangles exist on the split-bf16 kernel only
    const dim3 grid((unsigned)(pl.itiles * pl.jtiles), (unsigned)((smallc || pl.cols) ? 1 : s->R * s->S), (unsigned)pl.splitk);
    const bool incr = s->Wo >= BK;
#define RCF_WGRAD_LAUNCH(MRv, NRv)                                                                                \
    do {                                                                                                          \
        if (smallc) {                                                                                             \
            if (incr) hipLaunchKernelGGL((igemm_wgrad_kernel<MRv, NRv, true, true>), grid, dim3(256), 0, st, p);  \
            else hipLaunchKernelGGL((igemm_wgrad_kernel<MRv, NRv, false, true>), grid, dim3(256), 0, st, p);      \
        } else {                                                                                                  \
            if (incr) hipLaunchKernelGGL((igemm_wgrad_kernel<MRv, NRv, true, false>), grid, dim3(256), 0, st, p); \
            else hipLaunchKernelGGL((igemm_wgrad_kernel<MRv, NRv, false, false>), grid, dim3(256), 0, st, p);     \
        }                                                                                                         \
    } while (0)
    // narrow tiles: the fp32-MFMA kernel is as fast as the bf16 triples; with operand ranges they take the fp16 pairs
    // like every other conv of the step (same-box A/B: no difference in step time either way)
    const bool h2 = p.amax_a && p.amax_b;
    if (s->flags & (RCF_CONV_X_PLANES | RCF_CONV_DY_PLANES)) {
        // both operands as fp16 pair planes (igemm_h2dw.inc); one of them alone has no kernel
        const unsigned both = RCF_CONV_X_PLANES | RCF_CONV_DY_PLANES;
        if ((s->flags & both) != both || !h2 || !x3 || !pl.cols || pl.mr != 2 || s->Cin % 64 || s->Cout % 8 || s->x_pitch != s->Cin ||
            s->y_pitch != s->Cout)
            return RCF_EINVAL;
        if ((long)(pl.chunk / (long)p.rr + 2) * s->H * s->W * s->x_pitch * 4 >= (1L << 31)) return RCF_EINVAL;
        const bool onetap = s->Cin % (64 * pl.nr) == 0;
        p.cblocks = onetap && s->R * s->S > 1 && p.xcd_map ? s->Cin / (64 * pl.nr) : 0;
        if (pl.nr == 4) {
            if (region && onetap) hipLaunchKernelGGL((igemm_wgrad_h2d_kernel<4, true, true>), grid, dim3(256), 0, st, p);
            else if (region) hipLaunchKernelGGL((igemm_wgrad_h2d_kernel<4, true, false>), grid, dim3(256), 0, st, p);
            else if (onetap) hipLaunchKernelGGL((igemm_wgrad_h2d_kernel<4, false, true>), grid, dim3(256), 0, st, p);
            else hipLaunchKernelGGL((igemm_wgrad_h2d_kernel<4, false, false>), grid, dim3(256), 0, st, p);
        } else if (onetap) hipLaunchKernelGGL((igemm_wgrad_h2d_kernel<2, false, true>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((igemm_wgrad_h2d_kernel<2, false, false>), grid, dim3(256), 0, st, p);
    } else if (x3 && !smallc && ((pl.mr == 2 && pl.nr >= 2) || region || h2)) {
        if ((long)(pl.chunk / (long)p.rr + 2) * s->H * s->W * s->x_pitch * 4 >= (1L << 31)) return RCF_EINVAL;
        if (h2) {            // fp16 pairs
            if (pl.cols) {
                const bool onetap = s->Cin % (64 * pl.nr) == 0;
                p.cblocks = onetap && s->R * s->S > 1 && p.xcd_map ? s->Cin / (64 * pl.nr) : 0;
                if (pl.mr == 4) {
                    if (region) hipLaunchKernelGGL((igemm_wgrad_h2t_kernel<4, true, true, 4>), grid, dim3(256), 0, st, p);
                    else hipLaunchKernelGGL((igemm_wgrad_h2t_kernel<4, false, true, 4>), grid, dim3(256), 0, st, p);
                } else if (pl.nr == 4) {
                    if (region && onetap) hipLaunchKernelGGL((igemm_wgrad_h2t_kernel<4, true, true>), grid, dim3(256), 0, st, p);
                    else if (region) hipLaunchKernelGGL((igemm_wgrad_h2t_kernel<4, true, false>), grid, dim3(256), 0, st, p);
                    else if (onetap) hipLaunchKernelGGL((igemm_wgrad_h2t_kernel<4, false, true>), grid, dim3(256), 0, st, p);
                    else hipLaunchKernelGGL((igemm_wgrad_h2t_kernel<4, false, false>), grid, dim3(256), 0, st, p);
                } else if (onetap) hipLaunchKernelGGL((igemm_wgrad_h2t_kernel<2, false, true>), grid, dim3(256), 0, st, p);
                else hipLaunchKernelGGL((igemm_wgrad_h2t_kernel<2, false, false>), grid, dim3(256), 0, st, p);
            }
            else if (pl.nr == 4) hipLaunchKernelGGL(igemm_wgrad_x3_wide_kernel<2>, grid, dim3(256), 0, st, p);
            else if (pl.mr == 2 && pl.nr == 2) hipLaunchKernelGGL((igemm_wgrad_x3_kernel<2, 2, 2>), grid, dim3(256), 0, st, p);
            else if (pl.mr == 2 && pl.nr == 1) hipLaunchKernelGGL((igemm_wgrad_x3_kernel<2, 1, 2>), grid, dim3(256), 0, st, p);
            else if (pl.mr == 1 && pl.nr == 2) hipLaunchKernelGGL((igemm_wgrad_x3_kernel<1, 2, 2>), grid, dim3(256), 0, st, p);
            else hipLaunchKernelGGL((igemm_wgrad_x3_kernel<1, 1, 2>), grid, dim3(256), 0, st, p);
        } else if (pl.nr == 4) hipLaunchKernelGGL(igemm_wgrad_x3_wide_kernel<3>, grid, dim3(256), 0, st, p);
        else if (pl.mr == 2 && pl.nr == 2) hipLaunchKernelGGL((igemm_wgrad_x3_kernel<2, 2>), grid, dim3(256), 0, st, p);
        else if (pl.mr == 2 && pl.nr == 1) hipLaunchKernelGGL((igemm_wgrad_x3_kernel<2, 1>), grid, dim3(256), 0, st, p);
        else if (pl.mr == 1 && pl.nr == 2) hipLaunchKernelGGL((igemm_wgrad_x3_kernel<1, 2>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((igemm_wgrad_x3_kernel<1, 1>), grid, dim3(256), 0, st, p);
    } else if (pl.mr == 2 && pl.nr == 2) RCF_WGRAD_LAUNCH(2, 2);
    else if (pl.mr == 2 && pl.nr == 1) RCF_WGRAD_LAUNCH(2, 1);
    else if (pl.mr == 1 && pl.nr == 2) RCF_WGRAD_LAUNCH(1, 2);
    else RCF_WGRAD_LAUNCH(1, 1);
#undef RCF_WGRAD_LAUNCH
    RCF_LAUNCH_CHECK();
    if (pl.splitk > 1) {
        const long n4 = p.split_stride / 4;
        // small weights: one wavefront per workgroup, so that the few thousand float4 spread over all CUs
        const int bt = n4 < (1 << 17) ? 64 : 256;
        const int blocks = (int)((n4 + bt - 1) / bt < 4096 ? (n4 + bt - 1) / bt : 4096);
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(bt), 0, st, (const float *)workspace, dw, n4,
                           p.split_stride, pl.splitk, beta);
        RCF_LAUNCH_CHECK();
    }
    return 0;
}
