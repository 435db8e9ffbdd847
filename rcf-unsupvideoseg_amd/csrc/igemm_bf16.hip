// Convolution forward / data gradient / weight gradient with bf16 OPERANDS and fp32 accumulation -- the
// mixed-precision training step (BASELINE configs[2]: bf16 forward / fp32 gradients; the reference trains STv2 and
// FBMS under torch autocast, configs/rcf_stv2/rcf_stage1.yaml:57-60, configs/rcf_fbms59/rcf_stage1.yaml:61).
// Same implicit-GEMM view and reference call sites as igemm_conv.hip (models/resnet.py:164-203, models/res_layer.py:54-60,
// models/fcn_head.py:100-130); what changes is the storage: activations and activation gradients live in HBM as bf16
// NHWC (half the bytes of every pass), weights are cast once per launch from the fp32 master copy, weight gradients
// come out in fp32.  One MFMA pass (v_mfma_f32_32x32x16_bf16) instead of the three of the fp16-pair kernels: the
// roofline is the dense bf16 peak, 2.5 PF/s.
//
// K-step = 32 (64 B of one source pixel per GEMM row).  LDS tile [row][4 chunks of 16 B]; chunk c of row r is stored at
// chunk position c ^ ((r >> 2) & 3), which makes both the loader's ds_write_b128 and the MFMA operand fetch
// (ds_read_b128: lane = row, lane >> 5 = which half of a 16-k substep) bank-conflict free without padding.
#include <cstdlib>
#include <type_traits>
#include "rcf_common.h"

// No mutable process state: the A/B choices travel in rcf_conv_shape.flags (RCF_CONV_*), per call.

namespace {

typedef bf16_t bf16x8 __attribute__((ext_vector_type(8)));      // bf16_t: the build's 16-bit storage type (rcf_common.h)
typedef bf16_t bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr unsigned OOB = 0x80000000u;      // byte offset beyond every descriptor's num_records: loads return 0
constexpr int BKT = 32;                    // K-step (bf16 elements)

struct ConvParams {
    const bf16_t *A;      // source activations (x for fwd, dy for dgrad), NHWC bf16
    const bf16_t *Bw;     // weights, K-step major: [K/32][Ncol][32] (rcf_conv_weight_bf16)
    const float *bias;    // per output column or null
    void *Y;              // bf16 or fp32, NHWC
    int M, Ncol, K;
    int Ho, Wo;           // spatial dims of the GEMM-row tensor
    int Hs, Ws;           // spatial dims of the source tensor
    int Cs;               // source channels per tap (multiple of 8)
    int S;                // kernel width
    int up, off, step, div;   // source coord t = y*up + off + r*step, valid iff t>=0, t%div==0, t/div<Hs
    int a_pitch;
    long a_img_stride;
    int y_pitch;
    int act;
    float slope;
    int beta;
    int mtiles, ntiles;
    int mtiles8;                  // ceil(mtiles / 8): XCD x (block id % 8) walks the contiguous row tiles [x mtiles8, (x + 1) mtiles8)
    unsigned s_magic;
    int colmap;                   // rcf_common.h rcf_conv_tile: 1 = an XCD owns column tiles, not a band of row tiles
    int kch, rsch;                // K order (rcf_common.h rcf_kchunk): channel chunk width (Cs = natural order), taps * kch
    unsigned kch_magic, rsch_magic;
    int b_bytes;
    unsigned flags;                       // rcf_conv_shape.flags of the call
    int ry0, rx0, rh, rw, rband, rr;      // region of the GEMM-row tensor (see igemm_conv.hip)
    double *stats;                        // per row tile, fp64 column sums | sums of squares (forward; EP 2 data gradients)
    // fused epilogues of whole column tiles (template parameter EP of conv_bf16_kernel; bf16 output only):
    //   EP 1 (forward):        y = [max(0,] acc * ep_scale[c] + ep_shift[c] [+ ep_side[row][c]] [)]   -- conv -> batch norm (folded
    //                          into per-channel constants) -> residual add -> ReLU in the tile that computed the conv
    //   EP 2 (data gradient):  dx = ep_side[row][c] > 0 ? acc (+ old dx) : 0, column sums of what is written -> stats
    //                          (the ReLU of the join whose output this conv read, applied by the LAST writer of its gradient)
    //   EP 3                   the same with the mask from ep_bits (the forward tile's ballots) instead of the tensor
    const float *ep_scale, *ep_shift;
    const bf16_t *ep_side;
    int ep_side_pitch, ep_relu;
    // the ReLU's sign bits as the TILE decomposition produces them: one 64-bit ballot (lane = pixel row of the wave's 32-row tile
    // x channel half) per (wave, mr, nr, quad, element) -- 1 bit per output element, 1/16 of the bf16 tensor.  EP 1 with ReLU
    // writes them (ep_bits != null); EP 2 reads them INSTEAD of ep_side when given: both launches see the same [rows][Ncol]
    // tensor with the same tile shape, so a tile finds its own words back without any addressing by row or channel
    unsigned long long *ep_bits;
};

// v_cndmask with a wavefront-wide 64-bit mask held in scalar registers: out = mask[lane] ? v : 0
__device__ __forceinline__ float keep_if(float v, unsigned long long mask) {
    float out;
    asm volatile("v_cndmask_b32 %0, 0, %1, %2" : "=v"(out) : "v"(v), "s"(mask));
    return out;
}

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
// the same without a branch on the (uniform) divisor-is-one case: keeps a K-step's address arithmetic in one basic block
__device__ __forceinline__ int fast_div_nb(int k, unsigned magic) {
    return (int)(__umulhi((unsigned)k, magic) + ((unsigned)k & (magic ? 0u : ~0u)));
}

__device__ __forceinline__ u32x2 pack4(const f32x4 v) {
    const bf16x2 a = __builtin_convertvector(f32x2{v[0], v[1]}, bf16x2), b = __builtin_convertvector(f32x2{v[2], v[3]}, bf16x2);
    return u32x2{__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b)};
}

// one K-step of 32: two 16-k substeps of MR x NR MFMAs.  SWAP'd operands: the accumulator tile is transposed, a lane owns
// ONE tile row of A (one pixel) and, per register quad, four consecutive rows of B (output channels).
template <int MR, int NR>
__device__ __forceinline__ void mma_step(const char *__restrict__ As, const char *__restrict__ Bs, int arow0, int brow0,
                                         int lane, f32x16 (&acc)[MR][NR]) {
    const int l31 = lane & 31, kh = lane >> 5;
    const int sw = (l31 >> 2) & 3;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int coff = (((2 * j + kh) ^ sw) << 4);
        bf16x8 a[MR], b[NR];
#pragma unroll
        for (int mr = 0; mr < MR; ++mr) a[mr] = *reinterpret_cast<const bf16x8 *>(As + (arow0 + mr * 32 + l31) * 64 + coff);
#pragma unroll
        for (int nr = 0; nr < NR; ++nr) b[nr] = *reinterpret_cast<const bf16x8 *>(Bs + (brow0 + nr * 32 + l31) * 64 + coff);
#pragma unroll
        for (int mr = 0; mr < MR; ++mr)
#pragma unroll
            for (int nr = 0; nr < NR; ++nr)
                acc[mr][nr] = RCF_MFMA_32X32X16_H(b[nr], a[mr], acc[mr][nr]);
    }
}

// forward / data gradient.  Workgroup = WM x WN waves, each owning MR x NR accumulator tiles of 32x32.
// DGRAD only names the instantiation (profiles tell forward and data-gradient launches apart).
// OBF: bf16 output (16-byte stores after a half-wave exchange), else fp32 output.
// SCHED (DMA only): 1 = the steady-state K-step is one basic block with the LDS-DMA pieces placed among the MFMAs (the
// default), 0 = pieces issued ahead of the K-step's MFMAs (A/B reference, rcf_conv_bf16_set_tile(4)).
// DMA: both operands go from global memory STRAIGHT into LDS (buffer_load ... lds, 16 bytes per lane: no staging
// registers, no ds_write pass); three LDS stages, the loads of K-step t+2 are issued before the MFMAs of step t and
// stay in flight across the one barrier per step (counted s_waitcnt vmcnt).  The LDS image of a wave instruction is
// lane-linear (1 KB = 16 rows x 64 B), so the XOR swizzle of the 16-byte chunks is applied to the SOURCE address.
template <int MR, int NR, int WM, int WN, bool STRIDED, bool DGRAD, bool OBF, bool DMA = false, int NST = 3, int SCHED = 0, int EP = 0>
__global__ void __launch_bounds__(64 * WM * WN, (WM * WN == 4 && NST == 3) ? 2 : 1) conv_bf16_kernel(ConvParams p) {
    static_assert(EP == 0 || OBF, "the fused epilogues write bf16");
    constexpr int NT = 64 * WM * WN;
    constexpr int BM = 32 * MR * WM, BN = 32 * NR * WN;
    constexpr int PA = BM * 64, PB = BN * 64, STAGE = PA + PB;
    constexpr int NSTAGE = DMA ? NST : 2;
    __shared__ __attribute__((aligned(16))) char smem[NSTAGE * STAGE];

    // XCD aware (as igemm_conv_x3_kernel): ids b, b + 8, .. walk the column tiles of one row tile, each XCD its own contiguous
    // range of row tiles; p.colmap: each XCD its own column tiles of every row tile (rcf_common.h rcf_conv_tile)
    int tile_m, tile_n;
    rcf_conv_tile((int)blockIdx.x, p.mtiles8, p.ntiles, p.colmap, tile_m, tile_n);
    if (tile_m >= p.mtiles) return;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave - wm * WN;

    // Per-column constants of the straight-line epilogue (EP 1: scale | shift; otherwise the bias): asked for HERE, one column
    // per thread, and parked in LDS behind the K loop.  Read from global memory inside the epilogue they sit behind each
    // fragment's stores (a load may not pass a store it might alias): MR * NR dependent L2 round trips per tile.
    static_assert(BN <= NT, "one column constant per thread");
    static_assert(WM * BN * 2 * sizeof(double) + 2 * BN * sizeof(float) <= sizeof(smem), "statistics + column constants fit the stages");
    const bool use_cc = OBF && (EP == 1 || p.bias != nullptr) && n0 + BN <= p.Ncol;
    float cc0 = 0.f, cc1 = 0.f;
    if (use_cc && tid < BN) {
        if constexpr (EP == 1) { cc0 = p.ep_scale[n0 + tid]; cc1 = p.ep_shift[n0 + tid]; }
        else cc1 = p.bias[n0 + tid];
    }

    constexpr int ROWS = NT / 4;                      // 4 threads x 16 B per 64-byte row
    constexpr int A_PASS = BM / ROWS, B_PASS = BN / ROWS;
    static_assert(A_PASS >= 1 && B_PASS >= 1, "tile smaller than one loader pass");
    const int arow = tid >> 2;
    // register-staged path: this thread loads source chunk kq and stores it at position kq ^ swizzle; DMA path: the
    // thread's LDS position IS tid & 3, so it loads source chunk (tid & 3) ^ swizzle
    const int kq = DMA ? ((tid & 3) ^ ((arow >> 2) & 3)) : (tid & 3);
    const int st_off = arow * 64 + ((kq ^ ((arow >> 2) & 3)) << 4);      // + ROWS*64 per pass (ROWS % 16 == 0)

    const int HoWo = p.rr;
    const int n_first = m0 / HoWo;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16_t *>(p.A + (long)n_first * p.a_img_stride), 0, (int)OOB, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(p.Bw), 0, p.b_bytes, 0x00020000);

    int abase[A_PASS], ay[A_PASS], ax[A_PASS];
#pragma unroll
    for (int i = 0; i < A_PASS; ++i) {
        const int m = m0 + arow + ROWS * i;
        if (m < p.M) {
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
    unsigned bbase[B_PASS];   // byte offset of the weight row inside a K-step's block (out of range for columns past Ncol)
#pragma unroll
    for (int i = 0; i < B_PASS; ++i) {
        const int j = n0 + arow + ROWS * i;
        bbase[i] = j < p.Ncol ? (unsigned)j * 64u + (unsigned)kq * 16u : OOB;
    }
    f32x16 acc[MR][NR];
#pragma unroll
    for (int mr = 0; mr < MR; ++mr)
#pragma unroll
        for (int nr = 0; nr < NR; ++nr)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mr][nr][e] = 0.f;
    const int arow0 = wm * 32 * MR, brow0 = wn * 32 * NR;
    const int KT = (p.K + BKT - 1) / BKT;

    // byte offset of this thread's 16-byte chunk of A for K-step kt, row pass i (bit 31 set = out of range: zeros)
    auto a_offset = [&](int kt, int i, int tapoff, int c, bool kv, int dy, int dx) -> unsigned {
        int ty = ay[i] + dy, tx = ax[i] + dx;
        int v, off;
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
        return (((unsigned)off * 2u) & ~OOB) | ((unsigned)(v - 1) & OOB);
    };

    if constexpr (DMA) {
        // wave-uniform LDS destinations: pass i of wave w covers rows [ROWS i + 16 w, + 16) = 1 KB
        const int wrow = __builtin_amdgcn_readfirstlane(wave) * 16 * 64;
        auto issue = [&](int kt, int stage) {
            char *As = smem + stage * STAGE + wrow;
            char *Bs = As + PA;
            const int k = kt * BKT + kq * 8;
            const bool kv = k < p.K;
            // position k of the K loop = (channel chunk q, tap rs, channel inside the chunk); natural order: one chunk of Cs
            const int q = SCHED != 0 ? fast_div_nb(k, p.rsch_magic) : fast_div(k, p.rsch_magic);
            const int rem = k - q * p.rsch;
            const int rs = SCHED != 0 ? fast_div_nb(rem, p.kch_magic) : fast_div(rem, p.kch_magic);
            const int c = q * p.kch + (rem - rs * p.kch);
            const int r = SCHED != 0 ? fast_div_nb(rs, p.s_magic) : fast_div(rs, p.s_magic);
            const int s = rs - r * p.S;
            const int dy = r * p.step, dx = s * p.step;
            const int tapoff = (dy * p.Ws + dx) * p.a_pitch + c;
            const unsigned kb = (unsigned)kt * (unsigned)p.Ncol * 64u;
#pragma unroll
            for (int i = 0; i < B_PASS; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (__attribute__((address_space(3))) void *)(Bs + i * ROWS * 64), 16,
                                                         (int)(bbase[i] + kb), 0, 0, 0);
#pragma unroll
            for (int i = 0; i < A_PASS; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void *)(As + i * ROWS * 64), 16,
                                                         (int)a_offset(kt, i, tapoff, c, kv, dy, dx), 0, 0, 0);
        };
        constexpr int NLD = A_PASS + B_PASS;              // LDS-DMA instructions per thread per K-step
        constexpr int AHEAD = NST - 1;                    // K-steps in flight beyond the one being multiplied
        // wait until the loads of step `kt + 1` have landed: those of the later steps already issued stay in flight
        auto wait_next = [&](int kt) {
            const int later = min(KT - 1, kt + AHEAD) - (kt + 1);       // steps issued beyond kt + 1
            if (later >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NLD) : "memory");
            else if (later == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLD) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        };
#pragma unroll
        for (int i = 0; i < AHEAD; ++i)
            if (i < KT) issue(i, i);
        wait_next(-1);
        __builtin_amdgcn_s_barrier();
        int st = 0;                                       // stage of K-step kt
        if constexpr (SCHED != 0) {
            // steady state as ONE basic block (unconditional issue, constant wait count) so that the scheduler may place
            // the LDS-DMA pieces among the MFMAs; the last AHEAD steps, which issue nothing, follow
            int kt = 0;
            for (; kt + AHEAD < KT; ++kt) {
                const int stn = st + AHEAD >= NST ? st + AHEAD - NST : st + AHEAD;
                const char *As = smem + st * STAGE;
                mma_step<MR, NR>(As, As + PA, arow0, brow0, lane, acc);
                issue(kt + AHEAD, stn);       // in program order BEHIND the fragment reads (LDS write after LDS reads)
                {
                    // both halves' fragments first, then one LDS-DMA piece behind each of the first MFMAs: a piece's issue
                    // (tens of cycles) runs under the matrix pipe's work instead of ahead of it
                    __builtin_amdgcn_sched_group_barrier(0x100, 2 * (MR + NR), 0);
#pragma unroll
                    for (int i = 0; i < NLD; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, 2 * MR * NR - NLD, 0);
                }
                // the MFMAs may sink below the barrier (registers only); the fragment reads may not: the next step's
                // pieces overwrite this stage
                asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((AHEAD - 1) * NLD) : "memory");
                __builtin_amdgcn_s_barrier();
                st = st == NST - 1 ? 0 : st + 1;
            }
            for (; kt < KT; ++kt) {
                const char *As = smem + st * STAGE;
                mma_step<MR, NR>(As, As + PA, arow0, brow0, lane, acc);
                wait_next(kt);
                __builtin_amdgcn_s_barrier();
                st = st == NST - 1 ? 0 : st + 1;
            }
        } else {
        for (int kt = 0; kt < KT; ++kt) {
            // stage (st + AHEAD) % NST was read a step ago: free since the last barrier
            const int stn = st + AHEAD >= NST ? st + AHEAD - NST : st + AHEAD;
            if (kt + AHEAD < KT) issue(kt + AHEAD, stn);
            const char *As = smem + st * STAGE;
            mma_step<MR, NR>(As, As + PA, arow0, brow0, lane, acc);
            wait_next(kt);
            __builtin_amdgcn_s_barrier();
            st = st == NST - 1 ? 0 : st + 1;
        }
        }
    } else {
    // A (activations) comes from HBM: its loads run TWO K-steps ahead (two register sets); B (weights, L2) one step ahead
    u32x4 ra[2][A_PASS], rb[B_PASS];

    auto load_a = [&](int kt, u32x4 (&dst)[A_PASS]) {
        const int k = kt * BKT + kq * 8;
        const bool kv = k < p.K;
        const int q = fast_div(k, p.rsch_magic);
        const int rem = k - q * p.rsch;
        const int rs = fast_div(rem, p.kch_magic);
        const int c = q * p.kch + (rem - rs * p.kch);
        const int r = fast_div(rs, p.s_magic);
        const int s = rs - r * p.S;
        const int dy = r * p.step, dx = s * p.step;
        const int tapoff = (dy * p.Ws + dx) * p.a_pitch + c;
#pragma unroll
        for (int i = 0; i < A_PASS; ++i)
            dst[i] = __builtin_amdgcn_raw_buffer_load_b128(rsA, (int)a_offset(kt, i, tapoff, c, kv, dy, dx), 0, 0);
    };
    auto load_b = [&](int kt) {
        const unsigned kb = (unsigned)kt * (unsigned)p.Ncol * 64u;
#pragma unroll
        for (int i = 0; i < B_PASS; ++i) rb[i] = __builtin_amdgcn_raw_buffer_load_b128(rsB, (int)(bbase[i] + kb), 0, 0);
    };
    auto store_tile = [&](int buf, const u32x4 (&src)[A_PASS]) {
        char *As = smem + buf * STAGE;
        char *Bs = As + PA;
#pragma unroll
        for (int i = 0; i < A_PASS; ++i) *reinterpret_cast<u32x4 *>(As + st_off + i * ROWS * 64) = src[i];
#pragma unroll
        for (int i = 0; i < B_PASS; ++i) *reinterpret_cast<u32x4 *>(Bs + st_off + i * ROWS * 64) = rb[i];
    };

    load_a(0, ra[0]);
    load_b(0);
    load_a(1, ra[1]);                        // past the end of K: out-of-range offsets, zeros
    store_tile(0, ra[0]);
    __syncthreads();
    int kt = 0;
    for (; kt + 2 <= KT - 1; kt += 2) {
        {
            load_b(kt + 1);
            load_a(kt + 2, ra[0]);
            __builtin_amdgcn_sched_barrier(0);
            const char *As = smem;
            mma_step<MR, NR>(As, As + PA, arow0, brow0, lane, acc);
            store_tile(1, ra[1]);
            __syncthreads();
        }
        {
            load_b(kt + 2);
            load_a(kt + 3, ra[1]);
            __builtin_amdgcn_sched_barrier(0);
            const char *As = smem + STAGE;
            mma_step<MR, NR>(As, As + PA, arow0, brow0, lane, acc);
            store_tile(0, ra[0]);
            __syncthreads();
        }
    }
    if (kt < KT - 1) {
        load_b(kt + 1);
        __builtin_amdgcn_sched_barrier(0);
        const char *As = smem;
        mma_step<MR, NR>(As, As + PA, arow0, brow0, lane, acc);
        store_tile(1, ra[1]);
        __syncthreads();
        ++kt;
    }
    {
        const char *As = smem + (kt & 1) * STAGE;
        mma_step<MR, NR>(As, As + PA, arow0, brow0, lane, acc);
    }
    }

    // epilogue.  Transposed accumulators: lane = pixel (lane & 31) of each row tile, registers 4g..4g+3 = output channels
    // 8g + 4(lane>>5) .. +3 of each column tile.
    const int l31 = lane & 31, kh = lane >> 5;
    const bool full = p.rh == p.Ho && p.rw == p.Wo && p.rband <= 0;
    const bool want_stats = (!DGRAD || EP >= 2) && p.stats != nullptr;
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
    float *colc = reinterpret_cast<float *>(smem + WM * BN * 2 * sizeof(double));       // [2][BN] behind `red`
    if (want_stats || use_cc) __syncthreads();            // every wave is done with the operand stages
    if (use_cc) {
        if (tid < BN) { colc[tid] = cc0; colc[BN + tid] = cc1; }
        __syncthreads();
    }
    // The common case -- a whole column tile, no bias, no activation -- as straight-line code (csrc/igemm_conv.hip,
    // the same change there): the general loop tests columns / bias / activation / beta per quad and per element.
    // (a bias rides along as one uniform test per quad: the folded data gradient's second half, dx += x (-T) + c0)
    const bool lean = (p.bias == nullptr || (OBF && !want_stats)) && p.act == 0 && n0 + BN <= p.Ncol;
    const bool nts = (p.flags & RCF_CONV_NT_STORES) != 0;      // RCF_CONV_NT_STORES: the tile's output does not stay in L2
    // What the lean epilogue READS -- the residual / mask tensor (EP 1, 2), the ballots (EP 3), the old output (beta, bf16) --
    // is asked for ahead of the stores, two column tiles (2 MR fragments) at a time: a load of the output tensor may not pass
    // an earlier store to it, so loads left beside their fragment's stores run as MR * NR dependent HBM round trips
    // (a 256-deep 1x1 conv then spends 3 us multiplying and 15 us waiting; the whole tile's worth at once does not fit beside
    // its 128 accumulators).  16 bytes per lane in the STORE layout (a lane pair = 32 contiguous bytes of a row), brought into
    // the accumulator layout by the inverse half-wave exchange.
    constexpr bool PRE_SIDE = EP == 1 || EP == 2, PRE_BITS = EP == 3;
    constexpr int NRG = NR >= 4 ? 2 : 1;                  // column tiles asked for together
    u32x4 side_r[NRG][MR][2], old_r[NRG][MR][2];
    unsigned long long bits_r[NRG][MR];
    auto bits_index = [&](int mr, int nr) {
        return ((((long)tile_m * p.ntiles + tile_n) * (WM * WN) + wave) * (MR * NR) + (mr * NR + nr)) * 16;
    };
    auto prefetch_group = [&](int nr0) {
#pragma unroll
        for (int j = 0; j < NRG; ++j) {
            const int nr = nr0 + j;
            const int cb = n0 + brow0 + nr * 32;
#pragma unroll
            for (int mr = 0; mr < MR; ++mr) {
                if constexpr (PRE_SIDE) {
                    const bool rd = rowok[mr] && p.ep_side != nullptr;
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        side_r[j][mr][h] = u32x4{0u, 0u, 0u, 0u};
                        if (rd) side_r[j][mr][h] = *reinterpret_cast<const u32x4 *>(p.ep_side + lin[mr] * p.ep_side_pitch + cb + 8 * (2 * h + kh));
                    }
                }
                if constexpr (PRE_BITS) {
                    bits_r[j][mr] = 0ull;
                    if (lane < 16) bits_r[j][mr] = p.ep_bits[bits_index(mr, nr) + lane];
                }
                if constexpr (OBF && EP != 1) {           // (the affine entry point has no beta)
                    if (p.beta) {
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            old_r[j][mr][h] = u32x4{0u, 0u, 0u, 0u};
                            if (rowok[mr]) old_r[j][mr][h] = *reinterpret_cast<const u32x4 *>(reinterpret_cast<const bf16_t *>(p.Y) + lin[mr] * p.y_pitch + cb + 8 * (2 * h + kh));
                        }
                    }
                }
            }
        }
    };
    // 8 bf16 of channel groups (g, g + 1) in the store layout -> the two accumulator-layout quads of this lane
    auto unpack_pair = [&](const u32x4 d, f32x4 &lo, f32x4 &hi) {
        const auto s0 = __builtin_amdgcn_permlane32_swap(d[0], d[2], false, false);
        const auto s1 = __builtin_amdgcn_permlane32_swap(d[1], d[3], false, false);
        lo = rcf_widen4(s0[0], s1[0]);
        hi = rcf_widen4(s0[1], s1[1]);
    };
#pragma unroll
    for (int nr = 0; nr < NR; ++nr) {
        float cs[16], cq[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) { cs[e] = 0.f; cq[e] = 0.f; }
        const int cb = n0 + brow0 + nr * 32;              // first channel of this column tile
        if (lean) {
            if (nr % NRG == 0) prefetch_group(nr);
            const int nj = nr % NRG;
            auto quads = [&](auto BETA_, auto STATS_) {
                constexpr bool BETA = decltype(BETA_)::value, STATS = decltype(STATS_)::value;
#pragma unroll
                for (int mr = 0; mr < MR; ++mr) {
                    // rows past M multiplied zero activations: their accumulators are exactly zero
                    f32x4 q[4];
                    f32x4 side[4];                        // EP: the residual (1) / the activation whose sign masks the gradient (2)
                    f32x4 old[4];                         // beta, bf16 output: what the tile adds to
                    // EP 3: this (mr, nr) tile's 16 ballots, one per lane 0..15, handed out by readlane below
                    unsigned long long bitword = 0ull;
                    const long bits_at = bits_index(mr, nr);
                    if constexpr (EP == 3) bitword = bits_r[nj][mr];
                    unsigned long long mybits = 0ull;          // EP 1 with ReLU: the ballots this lane will store
                    if constexpr (EP == 1 || EP == 2) {
                        unpack_pair(side_r[nj][mr][0], side[0], side[1]);
                        unpack_pair(side_r[nj][mr][1], side[2], side[3]);
                    }
                    if constexpr (OBF && BETA) {
                        unpack_pair(old_r[nj][mr][0], old[0], old[1]);
                        unpack_pair(old_r[nj][mr][1], old[2], old[3]);
                    }
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int c = cb + 8 * g + 4 * kh;
                        f32x4 v = {acc[mr][nr][4 * g], acc[mr][nr][4 * g + 1], acc[mr][nr][4 * g + 2], acc[mr][nr][4 * g + 3]};
                        if (EP != 1 && p.bias != nullptr) v += *reinterpret_cast<const f32x4 *>(colc + BN + (c - n0));
                        if constexpr (EP == 1) {
                            v = v * *reinterpret_cast<const f32x4 *>(colc + (c - n0)) + *reinterpret_cast<const f32x4 *>(colc + BN + (c - n0));
                            v += side[g];                 // zeros without a residual
                            if (p.ep_relu) {
#pragma unroll
                                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                                if (p.ep_bits != nullptr) {
#pragma unroll
                                    for (int e = 0; e < 4; ++e) {
                                        const unsigned long long b = __ballot(v[e] > 0.f);
                                        if (lane == 4 * g + e) mybits = b;
                                    }
                                }
                            }
                        }
                        if constexpr (BETA) {
                            if constexpr (OBF) v += old[g];
                            else { if (rowok[mr]) v += ld4(reinterpret_cast<const float *>(p.Y) + lin[mr] * p.y_pitch + c); }
                        }
                        if constexpr (EP == 3) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const unsigned lo = __builtin_amdgcn_readlane((unsigned)bitword, 4 * g + e);
                                const unsigned hi = __builtin_amdgcn_readlane((unsigned)(bitword >> 32), 4 * g + e);
                                v[e] = keep_if(v[e], ((unsigned long long)hi << 32) | lo);
                            }
                        }
                        if constexpr (EP == 2) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = side[g][e] > 0.f ? v[e] : 0.f;
                        }
                        q[g] = v;
                        if (STATS) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                cs[4 * g + e] += v[e];
                                if constexpr (EP < 2) cq[4 * g + e] = fmaf(v[e], v[e], cq[4 * g + e]);     // (the masked gradient: sums only)
                            }
                        }
                    }
                    if constexpr (EP == 1) {
                        if (p.ep_relu && p.ep_bits != nullptr && lane < 16) p.ep_bits[bits_at + lane] = mybits;
                    }
                    if constexpr (OBF) {
                        bf16_t *yrow = reinterpret_cast<bf16_t *>(p.Y) + lin[mr] * p.y_pitch;
#pragma unroll
                        for (int g = 0; g < 4; g += 2) {
                            u32x2 a = pack4(q[g]), b = pack4(q[g + 1]);
                            const auto r0 = __builtin_amdgcn_permlane32_swap(a[0], b[0], false, false);
                            const auto r1 = __builtin_amdgcn_permlane32_swap(a[1], b[1], false, false);
                            if (rowok[mr]) {
                                u32x4 *dst = reinterpret_cast<u32x4 *>(yrow + cb + 8 * (g + kh));
                                if (nts) __builtin_nontemporal_store(u32x4{r0[0], r1[0], r0[1], r1[1]}, dst);
                                else *dst = u32x4{r0[0], r1[0], r0[1], r1[1]};
                            }
                        }
                    } else {
                        float *yrow = reinterpret_cast<float *>(p.Y) + lin[mr] * p.y_pitch;
                        if (rowok[mr]) {
#pragma unroll
                            for (int g = 0; g < 4; ++g) *reinterpret_cast<f32x4 *>(yrow + cb + 8 * g + 4 * kh) = q[g];
                        }
                    }
                }
            };
            if (EP != 1 && p.beta) { if (want_stats) quads(std::true_type{}, std::true_type{}); else quads(std::true_type{}, std::false_type{}); }
            else { if (want_stats) quads(std::false_type{}, std::true_type{}); else quads(std::false_type{}, std::false_type{}); }
        } else {
#pragma unroll
        for (int mr = 0; mr < MR; ++mr) {
            // lanes of rows past M keep zeros and take part in the half-wave exchange (no divergence around it)
            f32x4 q[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = cb + 8 * g + 4 * kh;
                f32x4 v = {acc[mr][nr][4 * g], acc[mr][nr][4 * g + 1], acc[mr][nr][4 * g + 2], acc[mr][nr][4 * g + 3]};
                const bool ok = rowok[mr] && c < p.Ncol;  // Ncol % 4 == 0: a quad is in or out as a whole
                if (ok) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (p.bias) v[e] += p.bias[c + e];
                        if (p.act == 1) v[e] = v[e] > 0.f ? v[e] : v[e] * p.slope;
                    }
                    if (p.beta) {
                        if constexpr (OBF) v += ld4(reinterpret_cast<const bf16_t *>(p.Y) + lin[mr] * p.y_pitch + c);
                        else v += ld4(reinterpret_cast<const float *>(p.Y) + lin[mr] * p.y_pitch + c);
                    }
                } else {
                    v = f32x4{0.f, 0.f, 0.f, 0.f};
                }
                q[g] = v;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    cs[4 * g + e] += v[e];
                    cq[4 * g + e] = fmaf(v[e], v[e], cq[4 * g + e]);
                }
            }
            if constexpr (OBF) {
                // half-wave exchange (v_permlane32_swap): lanes 0-31 end up with channels 8g..8g+7 of group pair (g, g+1)'s
                // first group, lanes 32-63 with those of the second -> one 16-byte store per pair
                bf16_t *yrow = reinterpret_cast<bf16_t *>(p.Y) + lin[mr] * p.y_pitch;
#pragma unroll
                for (int g = 0; g < 4; g += 2) {
                    u32x2 a = pack4(q[g]), b = pack4(q[g + 1]);
                    const auto r0 = __builtin_amdgcn_permlane32_swap(a[0], b[0], false, false);
                    const auto r1 = __builtin_amdgcn_permlane32_swap(a[1], b[1], false, false);
                    const int c8 = cb + 8 * (g + kh);
                    if (rowok[mr] && c8 < p.Ncol)
                        *reinterpret_cast<u32x4 *>(yrow + c8) = u32x4{r0[0], r1[0], r0[1], r1[1]};
                }
            } else {
                float *yrow = reinterpret_cast<float *>(p.Y) + lin[mr] * p.y_pitch;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int c = cb + 8 * g + 4 * kh;
                    if (rowok[mr] && c < p.Ncol) *reinterpret_cast<f32x4 *>(yrow + c) = q[g];
                }
            }
        }
        }
        if (want_stats) {                                 // block-uniform
            // column sums over the 32 pixels (lanes) of this half-wavefront: a reduce-scatter butterfly (igemm_conv.hip)
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
    if (want_stats) {
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

// fp32 master weights -> the bf16 operand of the kernels above, K-step major: element (row j, k) at
// ((k >> 5) * rows + j) * 32 + (k & 31), K padded with zeros to a multiple of 32.
// TRANSPOSE (data gradient): rows = c (Cin), k = rs * Cout + co of w[co][rs][c].
template <bool TRANSPOSE>
__global__ void __launch_bounds__(256) weight_bf16_kernel(const float *__restrict__ w, bf16_t *__restrict__ out, int Cout,
                                                          int Cin, int RS, int korder) {
    const int rows = TRANSPOSE ? Cin : Cout;
    const int K = TRANSPOSE ? RS * Cout : RS * Cin;
    const int Cs = TRANSPOSE ? Cout : Cin, kch = rcf_kchunk(korder, RS, Cs, RCF_KCHUNK_BF16);      // K order of the kernel that reads `out`
    const int KT = (K + 31) >> 5;
    const long n = (long)KT * rows * 32;
    const long step = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step) {
        const int kl = (int)(i & 31);
        const long t = i >> 5;
        const int j = (int)(t % rows);
        const int kp = (int)(t / rows) * 32 + kl;
        float v = 0.f;
        if (kp < K) {
            const int k = rcf_kperm(kp, RS, Cs, kch);
            if (!TRANSPOSE) {
                v = w[(long)j * K + k];
            } else {
                const int rs = k / Cout, co = k - rs * Cout;
                v = w[((long)co * RS + rs) * Cin + j];
            }
        }
        out[i] = (bf16_t)v;
    }
}

// ------------------------------------------------------------------------------------------- weight gradient
struct WgradParams {
    const bf16_t *X, *DY;
    float *OUT;
    int Cout, Cin, R, S;
    int H, W, Ho, Wo, stride, pad, dil;
    int x_pitch, dy_pitch;
    long M;            // N * rr
    long chunk;        // pixels per K-split (multiple of 32)
    int itiles, jtiles;
    long split_stride; // Cout*R*S*Cin
    int beta;          // only honoured when gridDim.z == 1
    int ry0, rx0, rh, rw, rband, rr;
    int Ktot;          // R*S*Cin: the GEMM columns are (tap, input channel) pairs
    int sched;         // LDS-DMA kernel: 1 = pieces interleaved with the MFMAs (default), 0 = ahead of them (A/B reference)
    int xcd_map;       // rcf_wgrad_item mode (1 = an XCD's workgroups share their pixels)
    int cblocks;       // rcf_wgrad_tile_ij (> 0: column tiles per tap, tiles numbered channel-block-major)
};

// byte offset `off` if ok == 1, an out-of-range offset (the load returns zeros) if ok == 0 -- arithmetic, so that a
// K-step's loads stay in one basic block
__device__ __forceinline__ unsigned oob_unless(unsigned off, int ok) { return (off & ~OOB) | ((unsigned)(ok - 1) & OOB); }

typedef short s16x4 __attribute__((__vector_size__(4 * sizeof(short))));
__device__ __forceinline__ u32x2 lds_tr16(const char *p) {
    const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4 *)(__attribute__((address_space(3))) char *)p);
    return __builtin_bit_cast(u32x2, v);
}

// dw[co][rs][c] = sum_m dy[m][co] * x[src(m, rs)][c]: rows i = co (64 MR), cols j = (rs, c) (64 NR), K = pixels (32 per
// step).  The taps are GEMM COLUMNS (j = rs * Cin + c, the weight's own memory order), not grid rows: a tile of a
// narrow layer (Cin = 64: 576 columns) spans several taps, so dy is read once per 256 columns instead of once per tap.
// Both operands are staged in their NATURAL order -- LDS holds [32 pixels][channels] bf16 (16-byte global loads of 8
// channels of a pixel, one ds_write_b128 each) -- and the MFMA's k-contiguous fragments come out of gfx950's transposing
// LDS read (ds_read_b64_tr_b16), as in igemm_wgrad_h2t_kernel.
// REGION: the contributing output pixels are a rectangle / frame of every image (general pixel walk).
template <int MR, int NR, bool REGION>
__global__ void __launch_bounds__(256, 2) wgrad_bf16_kernel(WgradParams p) {
    constexpr int BM = 64 * MR, BN = 64 * NR, BK = 32;
    constexpr int QA = BM / 8, PAS = 256 / QA, NAP = BK / PAS;   // dy loader: 16-byte chunks per pixel, pixels per pass, passes
    constexpr int QB = BN / 8, PBS = 256 / QB, NBP = BK / PBS;   // x loader
    // row pitch = channels * 2 + 64 bytes: the 4 pixel rows a 16-lane group reads are 64 B apart in bank space and the
    // second group of a 32-lane half (channels + 16 = + 32 B) falls into the gaps: 256 distinct bytes per LDS cycle
    constexpr int PIA = BM * 2 + 64, PIB = BN * 2 + 64;
    constexpr int PLA = BK * PIA, PLB = BK * PIB, STAGE = PLA + PLB;
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];

    int tile, split;
    rcf_wgrad_item(p.xcd_map, tile, split);
    int tile_i, tile_j;
    rcf_wgrad_tile_ij(tile, p.itiles, p.jtiles, p.cblocks, tile_i, tile_j);
    const int i0 = tile_i * BM, j0 = tile_j * BN;
    const long kbeg = (long)split * p.chunk;
    const long kend = min(p.M, kbeg + p.chunk);
    const int klen = (int)(kend - kbeg);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int HoWo = p.rr;
    const int n_first = (int)(kbeg / HoWo);
    // whole tensors: the contributing pixels are the rows of dy in order, no walk needed
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16_t *>(p.DY + (REGION ? (long)n_first * p.Ho * p.Wo : kbeg) * p.dy_pitch), 0, (int)OOB, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16_t *>(p.X + (long)n_first * p.H * p.W * p.x_pitch), 0, (int)OOB, 0x00020000);

    const int qa = tid % QA, pa0 = tid / QA;
    const int qb = tid % QB, pb0 = tid / QB;
    const int cha = i0 + 8 * qa, jc = j0 + 8 * qb;          // this thread's dy channels / GEMM columns (one tap: Cin % 8 == 0)
    const int rs = jc / p.Cin, chb = jc - rs * p.Cin;
    const int r = rs / p.S, s = rs - r * p.S;
    const bool acta = cha < p.Cout, actb = jc < p.Ktot;
    // pixel walkers (image-relative to n_first), advanced by BK per K-step
    int an[NAP], ay[NAP], ax[NAP], apix[NAP];
    int bn[NBP], by[NBP], bx[NBP], bpix[NBP];
    auto init_px = [&](long m, int &n, int &y, int &x, int &pix) {
        n = (int)(m / HoWo);
        pix = (int)(m - (long)n * HoWo);
        if (REGION) region_yx(pix, p.ry0, p.rx0, p.rh, p.rw, p.rband, y, x);
        else { y = pix / p.Wo; x = pix - y * p.Wo; }
        n -= n_first;
    };
    if (REGION) {
#pragma unroll
        for (int j = 0; j < NAP; ++j) init_px(kbeg + pa0 + PAS * j, an[j], ay[j], ax[j], apix[j]);
    }
#pragma unroll
    for (int j = 0; j < NBP; ++j) init_px(kbeg + pb0 + PBS * j, bn[j], by[j], bx[j], bpix[j]);
    const bool incr = !REGION && p.Wo >= BK;
    auto advance = [&](int &n, int &y, int &x, int &pix) {
        if (incr) {                              // Wo >= BK: at most one row wrap per K-step
            x += BK;
            const bool wx = x >= p.Wo;
            x -= wx ? p.Wo : 0;
            y += wx ? 1 : 0;
            const bool wy = y == p.Ho;
            y = wy ? 0 : y;
            n += wy ? 1 : 0;
        } else {
            pix += BK;
            while (pix >= HoWo) { pix -= HoWo; ++n; }
            if (REGION) region_yx(pix, p.ry0, p.rx0, p.rh, p.rw, p.rband, y, x);
            else { y = pix / p.Wo; x = pix - y * p.Wo; }
        }
    };

    u32x4 ra[NAP], rb[NBP];
    auto load_tile = [&](int kt) {
#pragma unroll
        for (int j = 0; j < NAP; ++j) {
            const int mk = kt * BK + pa0 + PAS * j;
            const bool v = acta && mk < klen;
            unsigned bo;
            if (REGION) {
                bo = v ? (unsigned)(((an[j] * p.Ho + ay[j]) * p.Wo + ax[j]) * p.dy_pitch + cha) * 2u : OOB;
                advance(an[j], ay[j], ax[j], apix[j]);
            } else {
                bo = v ? (unsigned)(mk * p.dy_pitch + cha) * 2u : OOB;
            }
            ra[j] = __builtin_amdgcn_raw_buffer_load_b128(rsA, (int)bo, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < NBP; ++j) {
            const int mk = kt * BK + pb0 + PBS * j;
            const int sy = by[j] * p.stride - p.pad + r * p.dil;
            const int sx = bx[j] * p.stride - p.pad + s * p.dil;
            const bool v = actb && mk < klen && (unsigned)sy < (unsigned)p.H && (unsigned)sx < (unsigned)p.W;
            const unsigned bo = v ? (unsigned)(((bn[j] * p.H + sy) * p.W + sx) * p.x_pitch + chb) * 2u : OOB;
            rb[j] = __builtin_amdgcn_raw_buffer_load_b128(rsB, (int)bo, 0, 0);
            advance(bn[j], by[j], bx[j], bpix[j]);
        }
    };
    auto store_tile = [&](int buf) {
        char *As = smem + buf * STAGE, *Bs = As + PLA;
#pragma unroll
        for (int j = 0; j < NAP; ++j) *reinterpret_cast<u32x4 *>(As + (pa0 + PAS * j) * PIA + qa * 16) = ra[j];
#pragma unroll
        for (int j = 0; j < NBP; ++j) *reinterpret_cast<u32x4 *>(Bs + (pb0 + PBS * j) * PIB + qb * 16) = rb[j];
    };

    f32x16 acc[MR][NR];
#pragma unroll
    for (int mr = 0; mr < MR; ++mr)
#pragma unroll
        for (int nr = 0; nr < NR; ++nr)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mr][nr][e] = 0.f;

    const int arow0 = wm * 32 * MR, brow0 = wn * 32 * NR;
    // fragment addressing (see igemm_wgrad_h2t_kernel): 16-lane group gi serves tile rows 16 (gi & 1) .. +15 and the k
    // half gi >> 1; lane q of the group points at pixel row 8 (gi >> 1) + q / 4 (+ 4 for the second read), channels 4 (q % 4)
    const int gi = lane >> 4, q16 = lane & 15;
    const int fa = (8 * (gi >> 1) + (q16 >> 2)) * PIA + (arow0 + 16 * (gi & 1) + 4 * (q16 & 3)) * 2;
    const int fb = (8 * (gi >> 1) + (q16 >> 2)) * PIB + (brow0 + 16 * (gi & 1) + 4 * (q16 & 3)) * 2;
    auto mma = [&](int buf) {
        const char *As = smem + buf * STAGE, *Bs = As + PLA;
#pragma unroll
        for (int j = 0; j < 2; ++j) {                     // two 16-pixel substeps
            bf16x8 a[MR], b[NR];
#pragma unroll
            for (int mr = 0; mr < MR; ++mr) {
                const u32x2 lo = lds_tr16(As + fa + 16 * j * PIA + mr * 64);
                const u32x2 hi = lds_tr16(As + fa + 16 * j * PIA + mr * 64 + 4 * PIA);
                a[mr] = __builtin_bit_cast(bf16x8, u32x4{lo[0], lo[1], hi[0], hi[1]});
            }
#pragma unroll
            for (int nr = 0; nr < NR; ++nr) {
                const u32x2 lo = lds_tr16(Bs + fb + 16 * j * PIB + nr * 64);
                const u32x2 hi = lds_tr16(Bs + fb + 16 * j * PIB + nr * 64 + 4 * PIB);
                b[nr] = __builtin_bit_cast(bf16x8, u32x4{lo[0], lo[1], hi[0], hi[1]});
            }
#pragma unroll
            for (int mr = 0; mr < MR; ++mr)
#pragma unroll
                for (int nr = 0; nr < NR; ++nr)
                    acc[mr][nr] = RCF_MFMA_32X32X16_H(a[mr], b[nr], acc[mr][nr]);
        }
    };

    const int KT = (klen + BK - 1) / BK;
    load_tile(0);
    store_tile(0);
    __syncthreads();
    for (int kt = 0; kt + 1 < KT; ++kt) {
        const int cur = kt & 1;
        load_tile(kt + 1);
        __builtin_amdgcn_sched_barrier(0);
        mma(cur);
        store_tile(cur ^ 1);
        __syncthreads();
    }
    mma((KT - 1) & 1);

    float *out = p.OUT + (long)split * p.split_stride;
    const int l31 = lane & 31, kh = lane >> 5;
    if (i0 + 32 * MR * 2 <= p.Cout && j0 + 32 * NR * 2 <= p.Ktot && !(p.beta && gridDim.z == 1)) {
        // the tile lies inside the weight tensor and is a split-K partial (or overwrites): store, nothing to test
#pragma unroll
        for (int mr = 0; mr < MR; ++mr)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float *drow = out + (long)(i0 + arow0 + mr * 32 + 4 * kh + (e & 3) + 8 * (e >> 2)) * p.Ktot + j0 + brow0 + l31;
#pragma unroll
                for (int nr = 0; nr < NR; ++nr) drow[nr * 32] = acc[mr][nr][e];
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
            float *drow = out + (long)co * p.Ktot + j0 + brow0 + l31;
#pragma unroll
            for (int nr = 0; nr < NR; ++nr) {
                if (j0 + brow0 + nr * 32 + l31 >= p.Ktot) continue;
                float v = acc[mr][nr][e];
                if (p.beta && gridDim.z == 1) v += drow[nr * 32];
                drow[nr * 32] = v;
            }
        }
    }
}

// one K-step of the LDS-DMA weight-gradient kernel: two 16-pixel substeps, fragments through the transposing reads.
// `As` / `Bs` are __restrict__ on purpose: inlined, the reads carry no-alias scopes against the kernel's LDS-DMA writes --
// without them the compiler puts `s_waitcnt vmcnt(0)` in front of the first read of every K-step (it must assume the
// pieces in flight, which target ANOTHER stage, could alias) and the loads of later steps no longer stay in flight.
template <int MR, int NR>
__device__ __forceinline__ void wgrad_tr_step(const char *__restrict__ As, const char *__restrict__ Bs, const int (&fa)[MR],
                                              const int (&fb)[NR], f32x16 (&acc)[MR][NR]) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        bf16x8 a[MR], b[NR];
#pragma unroll
        for (int mr = 0; mr < MR; ++mr) {
            const u32x2 lo = lds_tr16(As + fa[mr] + 16 * j * 128);
            const u32x2 hi = lds_tr16(As + fa[mr] + 16 * j * 128 + 4 * 128);
            a[mr] = __builtin_bit_cast(bf16x8, u32x4{lo[0], lo[1], hi[0], hi[1]});
        }
#pragma unroll
        for (int nr = 0; nr < NR; ++nr) {
            const u32x2 lo = lds_tr16(Bs + fb[nr] + 16 * j * 128);
            const u32x2 hi = lds_tr16(Bs + fb[nr] + 16 * j * 128 + 4 * 128);
            b[nr] = __builtin_bit_cast(bf16x8, u32x4{lo[0], lo[1], hi[0], hi[1]});
        }
#pragma unroll
        for (int mr = 0; mr < MR; ++mr)
#pragma unroll
            for (int nr = 0; nr < NR; ++nr)
                acc[mr][nr] = RCF_MFMA_32X32X16_H(a[mr], b[nr], acc[mr][nr]);
    }
}

// The 128-row tiles again with LDS-DMA loads (buffer_load ... lds), three LDS stages and one barrier per K-step, like
// conv_bf16_kernel<..., DMA>.  LDS image per operand: [channel group of 64][32 pixels][128 B] -- a DMA instruction (1 KB,
// lane-linear) is 8 pixel rows of ONE channel group, so every thread serves ONE pixel (8 wave + lane / 8) in all of its
// instructions: one pixel walk and one bounds test per K-step instead of one per load (the register-staged kernel spends
// 9.5 VALU instructions per MFMA, mostly on that).  The 16-byte chunks of pixel row p are XOR-swizzled by
// ((p >> 1) & 1) << 2 (source address and fragment address alike): the 4 rows x 4 chunks a 32-lane half reads through
// ds_read_b64_tr_b16 cover all 16 chunk positions of the 256-byte bank space (SQ_LDS_BANK_CONFLICT = 0).
// ONETAP: Cin is a multiple of the tile width, so a tile's columns belong to one filter tap.
// MR = 4 (rcf_conv_set_wgrad_big): 256 x 256 tile, one workgroup per CU (96 KB of LDS, 256 accumulator registers per wave).
template <int NR, bool REGION, bool ONETAP, int MR = 2>
__global__ void __launch_bounds__(256, MR == 2 ? 2 : 1) wgrad_bf16_dma_kernel(WgradParams p) {
    constexpr int BM = 64 * MR, BN = 64 * NR, BK = 32;
    constexpr int GA = BM / 64, GB = BN / 64;                  // channel groups = DMA instructions per wave and K-step
    constexpr int GSZ = BK * 128;                              // bytes of one group: 32 pixels x 128 B
    constexpr int PLA = GA * GSZ, PLB = GB * GSZ, STAGE = PLA + PLB;
    __shared__ __attribute__((aligned(16))) char smem[3 * STAGE];

    int tile, split;
    rcf_wgrad_item(p.xcd_map, tile, split);
    int tile_i, tile_j;
    rcf_wgrad_tile_ij(tile, p.itiles, p.jtiles, p.cblocks, tile_i, tile_j);
    const int i0 = tile_i * BM, j0 = tile_j * BN;
    const long kbeg = (long)split * p.chunk;
    const long kend = min(p.M, kbeg + p.chunk);
    const int klen = (int)(kend - kbeg);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int HoWo = p.rr;
    const int n_first = (int)(kbeg / HoWo);
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16_t *>(p.DY + (REGION ? (long)n_first * p.Ho * p.Wo : kbeg) * p.dy_pitch), 0, (int)OOB, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16_t *>(p.X + (long)n_first * p.H * p.W * p.x_pitch), 0, (int)OOB, 0x00020000);

    // this thread's pixel row of the K-step and its chunk position inside a 128-byte group row
    const int prow = 8 * wave + (lane >> 3), qpos = lane & 7;
    const int qsrc = qpos ^ (((prow >> 1) & 1) << 2);          // source chunk of that position
    int pn, py, px, ppix;                                      // the pixel walk (image index relative to n_first)
    {
        const long m = kbeg + prow;
        pn = (int)(m / HoWo);
        ppix = (int)(m - (long)pn * HoWo);
        if (REGION) region_yx(ppix, p.ry0, p.rx0, p.rh, p.rw, p.rband, py, px);
        else { py = ppix / p.Wo; px = ppix - py * p.Wo; }
        pn -= n_first;
    }
    const bool incr = !REGION && p.Wo >= BK;
    int cha[GA], chb[GB], tr[GB], ts[GB];
    bool acta[GA], actb[GB];
#pragma unroll
    for (int g = 0; g < GA; ++g) {
        cha[g] = i0 + 64 * g + 8 * qsrc;
        acta[g] = cha[g] < p.Cout;
    }
#pragma unroll
    for (int g = 0; g < GB; ++g) {
        const int jc = j0 + 64 * g + 8 * qsrc;                 // GEMM column = (tap, channel)
        const int rs = (ONETAP ? j0 : jc) / p.Cin;
        chb[g] = jc - rs * p.Cin;
        tr[g] = rs / p.S;
        ts[g] = rs - tr[g] * p.S;
        actb[g] = jc < p.Ktot;
    }
    const int wb = __builtin_amdgcn_readfirstlane(wave) * 1024;    // 8 pixel rows x 128 B of each group
    auto loads = [&](int kt, int stage) {                  // K-steps are issued in order: the walk advances by BK each time
        char *As = smem + stage * STAGE + wb;
        char *Bs = smem + stage * STAGE + PLA + wb;
        const int mk = kt * BK + prow;
        const bool inb = mk < klen;
        const int dyoff = REGION ? ((pn * p.Ho + py) * p.Wo + px) * p.dy_pitch : mk * p.dy_pitch;
#pragma unroll
        for (int g = 0; g < GA; ++g) {
            const unsigned bo = oob_unless((unsigned)(dyoff + cha[g]) * 2u, (int)inb & (int)acta[g]);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void *)(As + g * GSZ), 16, (int)bo, 0, 0, 0);
        }
        if constexpr (ONETAP) {
            const int sy = py * p.stride - p.pad + tr[0] * p.dil, sx = px * p.stride - p.pad + ts[0] * p.dil;
            const int v = (int)inb & (int)((unsigned)sy < (unsigned)p.H) & (int)((unsigned)sx < (unsigned)p.W);
            const int xoff = ((pn * p.H + sy) * p.W + sx) * p.x_pitch;
#pragma unroll
            for (int g = 0; g < GB; ++g) {
                const unsigned bo = oob_unless((unsigned)(xoff + chb[g]) * 2u, (int)v & (int)actb[g]);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (__attribute__((address_space(3))) void *)(Bs + g * GSZ), 16, (int)bo, 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int g = 0; g < GB; ++g) {
                const int sy = py * p.stride - p.pad + tr[g] * p.dil, sx = px * p.stride - p.pad + ts[g] * p.dil;
                const int v = (int)inb & (int)actb[g] & (int)((unsigned)sy < (unsigned)p.H) & (int)((unsigned)sx < (unsigned)p.W);
                const unsigned bo = oob_unless((unsigned)(((pn * p.H + sy) * p.W + sx) * p.x_pitch + chb[g]) * 2u, v);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (__attribute__((address_space(3))) void *)(Bs + g * GSZ), 16, (int)bo, 0, 0, 0);
            }
        }
    };
    auto advance_rows = [&]() {                  // Wo >= BK: at most one row wrap per K-step (no branches)
        px += BK;
        const bool wx = px >= p.Wo;
        px -= wx ? p.Wo : 0;
        py += wx ? 1 : 0;
        const bool wy = py == p.Ho;
        py = wy ? 0 : py;
        pn += wy ? 1 : 0;
    };
    auto issue = [&](int kt, int stage) {
        loads(kt, stage);
        if (incr) {
            advance_rows();
        } else {
            ppix += BK;
            while (ppix >= HoWo) { ppix -= HoWo; ++pn; }
            if (REGION) region_yx(ppix, p.ry0, p.rx0, p.rh, p.rw, p.rband, py, px);
            else { py = ppix / p.Wo; px = ppix - py * p.Wo; }
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
    // fragment addressing: 16-lane group gi serves tile rows (channels) 16 (gi & 1) .. +15 and the k half gi >> 1; lane q of
    // the group points at pixel row 8 (gi >> 1) + q / 4 (+ 4 for the second read), channels 4 (q % 4) .. + 3 of its 16
    const int gi = lane >> 4, q16 = lane & 15;
    const int fprow = 8 * (gi >> 1) + (q16 >> 2);
    const int swz = ((fprow >> 1) & 1) << 2;                              // (+4, +16 j leave bit 1 of the pixel row alone)
    const int cin = 16 * (gi & 1) + 4 * (q16 & 3);                         // channel inside a 32-channel tile
    const int wi = (cin & 7) * 2;
    int fa[MR], fb[NR];
#pragma unroll
    for (int mr = 0; mr < MR; ++mr) {
        const int c = arow0 + mr * 32 + cin;
        fa[mr] = (c >> 6) * GSZ + fprow * 128 + ((((c >> 3) & 7) ^ swz) << 4) + wi;
    }
#pragma unroll
    for (int nr = 0; nr < NR; ++nr) {
        const int c = brow0 + nr * 32 + cin;
        fb[nr] = (c >> 6) * GSZ + fprow * 128 + ((((c >> 3) & 7) ^ swz) << 4) + wi;
    }
    auto mma = [&](int stage) {
        const char *As = smem + stage * STAGE;
        wgrad_tr_step<MR, NR>(As, As + PLA, fa, fb, acc);
    };

    constexpr int NLD = GA + GB;
    const int KT = (klen + BK - 1) / BK;
    issue(0, 0);
    if (KT > 1) issue(1, 1);
    if (KT > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLD) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    int st = 0, kt = 0;
    if (incr && p.sched) {
        // steady state as ONE basic block (conv_bf16_kernel<..., SCHED = 1>): fragment reads, then one LDS-DMA piece behind
        // each of the first MFMAs
        for (; kt + 2 < KT; ++kt) {
            const int st2 = st >= 1 ? st - 1 : 2;         // (st + 2) % 3
            mma(st);
            loads(kt + 2, st2);
            advance_rows();
            if constexpr (MR == 4) {
                // 256 x 256: the fragments of ONE k half at a time (64 registers for both would spill beside 256 accumulators)
                __builtin_amdgcn_sched_group_barrier(0x100, 2 * (MR + NR), 0);
#pragma unroll
                for (int i = 0; i < NLD; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, MR * NR - NLD, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 2 * (MR + NR), 0);
                __builtin_amdgcn_sched_group_barrier(0x008, MR * NR, 0);
            } else {
            __builtin_amdgcn_sched_group_barrier(0x100, 4 * (MR + NR), 0);
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 2 * MR * NR - NLD, 0);
            }
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NLD) : "memory");
            __builtin_amdgcn_s_barrier();
            st = st == 2 ? 0 : st + 1;
        }
    }
    for (; kt < KT; ++kt) {
        const int st2 = st >= 1 ? st - 1 : 2;
        if (kt + 2 < KT) issue(kt + 2, st2);
        mma(st);
        if (kt + 2 < KT) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLD) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        st = st == 2 ? 0 : st + 1;
    }

    float *out = p.OUT + (long)split * p.split_stride;
    const int l31 = lane & 31, kh = lane >> 5;
    if (i0 + 32 * MR * 2 <= p.Cout && j0 + 32 * NR * 2 <= p.Ktot && !(p.beta && gridDim.z == 1)) {
        // the tile lies inside the weight tensor and is a split-K partial (or overwrites): store, nothing to test
#pragma unroll
        for (int mr = 0; mr < MR; ++mr)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float *drow = out + (long)(i0 + arow0 + mr * 32 + 4 * kh + (e & 3) + 8 * (e >> 2)) * p.Ktot + j0 + brow0 + l31;
#pragma unroll
                for (int nr = 0; nr < NR; ++nr) drow[nr * 32] = acc[mr][nr][e];
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
            float *drow = out + (long)co * p.Ktot + j0 + brow0 + l31;
#pragma unroll
            for (int nr = 0; nr < NR; ++nr) {
                if (j0 + brow0 + nr * 32 + l31 >= p.Ktot) continue;
                float v = acc[mr][nr][e];
                if (p.beta && gridDim.z == 1) v += drow[nr * 32];
                drow[nr * 32] = v;
            }
        }
    }
}

__global__ void splitk_reduce_kernel(const float *__restrict__ ws, float *__restrict__ dw, long n4, long stride,
                                     int splits, int beta) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long step = (long)gridDim.x * blockDim.x;
    for (; i < n4; i += step) {
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

// ------------------------------------------------------------------------------------------- host side
// cout_mult: 8 where Cout is read or written as bf16 (16-byte accesses), 4 for the forward with fp32 output
int check_shape(const rcf_conv_shape *s, int cout_mult = 8) {
    if (!s || s->struct_bytes != sizeof(rcf_conv_shape)) return RCF_EINVAL;      // a caller built against another header
    if (s->N <= 0 || s->H <= 0 || s->W <= 0 || s->Cin <= 0 || s->Cout <= 0 || s->R <= 0 || s->S <= 0) return RCF_EINVAL;
    // 16-byte loads of 8 bf16 channels: channel counts and pitches in multiples of 8
    if (s->Cin % 8 || s->Cout % cout_mult || s->x_pitch % 8 || s->x_pitch < s->Cin || s->y_pitch < s->Cout) return RCF_EINVAL;
    if (s->stride <= 0 || s->dil <= 0 || s->pad < 0) return RCF_EINVAL;
    const int ho = (s->H + 2 * s->pad - s->dil * (s->R - 1) - 1) / s->stride + 1;
    const int wo = (s->W + 2 * s->pad - s->dil * (s->S - 1) - 1) / s->stride + 1;
    if (ho != s->Ho || wo != s->Wo) return RCF_EINVAL;
    if ((long)s->N * s->Ho * s->Wo >= (1L << 31) || (long)s->N * s->H * s->W >= (1L << 31)) return RCF_EINVAL;
    return 0;
}

inline unsigned magic_of(int d) { return d <= 1 ? 0u : (unsigned)((1ull << 32) / (unsigned)d + 1ull); }

inline int region_pixels(const rcf_conv_region *r, int H, int W) {
    if (!r) return H * W;
    return r->band > 0 ? 2 * r->band * r->w + 2 * r->band * (r->h - 2 * r->band) : r->h * r->w;
}

int set_region(ConvParams &p, const rcf_conv_region *r, int N, int H, int W) {
    p.ry0 = r ? r->y0 : 0; p.rx0 = r ? r->x0 : 0; p.rh = r ? r->h : H; p.rw = r ? r->w : W;
    p.rband = r ? r->band : 0;
    if (p.ry0 < 0 || p.rx0 < 0 || p.rh <= 0 || p.rw <= 0 || p.ry0 + p.rh > H || p.rx0 + p.rw > W) return RCF_EINVAL;
    if (p.rband < 0 || (p.rband > 0 && (2 * p.rband >= p.rh || 2 * p.rband >= p.rw))) return RCF_EINVAL;
    p.rr = region_pixels(r, H, W);
    p.M = N * p.rr;
    return 0;
}

inline bool korder_chunked(unsigned flags) { return !(flags & RCF_CONV_KORDER_NATURAL); }

// LDS-DMA kernels (128x64 / 128x128 / 128x256 by output width), three stages, pieces interleaved with the MFMAs
template <int MR, int NR, int WM, int WN, bool OBF>
void launch_cfg(ConvParams &p, bool strided, bool dgrad, hipStream_t st, int ep) {
    constexpr int BM = 32 * MR * WM, BN = 32 * NR * WN;
    p.mtiles = rcf_cdiv(p.M, BM);
    p.mtiles8 = rcf_cdiv(p.mtiles, 8);
    p.ntiles = rcf_cdiv(p.Ncol, BN);
    p.colmap = p.Ncol % BN == 0 && rcf_colmap_pays(!(p.flags & RCF_CONV_NO_COLMAP), (long)p.M * p.Cs * 2, (long)p.K * p.Ncol * 2, p.mtiles, p.ntiles);
    const dim3 grid((unsigned)(rcf_cdiv(p.mtiles, 8) * 8 * p.ntiles));
    if constexpr (OBF) {
        // fused epilogues (the caller checked: whole column tiles, stride 1, no bias / activation)
        if (ep == 1) { hipLaunchKernelGGL((conv_bf16_kernel<MR, NR, WM, WN, false, false, true, true, 3, 1, 1>), grid, dim3(64 * WM * WN), 0, st, p); return; }
        if (ep == 2 && p.ep_bits != nullptr) { hipLaunchKernelGGL((conv_bf16_kernel<MR, NR, WM, WN, false, true, true, true, 3, 1, 3>), grid, dim3(64 * WM * WN), 0, st, p); return; }
        if (ep == 2) { hipLaunchKernelGGL((conv_bf16_kernel<MR, NR, WM, WN, false, true, true, true, 3, 1, 2>), grid, dim3(64 * WM * WN), 0, st, p); return; }
    }
    if (strided) hipLaunchKernelGGL((conv_bf16_kernel<MR, NR, WM, WN, true, true, OBF, true, 3, 1>), grid, dim3(64 * WM * WN), 0, st, p);
    else if (dgrad) hipLaunchKernelGGL((conv_bf16_kernel<MR, NR, WM, WN, false, true, OBF, true, 3, 1>), grid, dim3(64 * WM * WN), 0, st, p);
    else hipLaunchKernelGGL((conv_bf16_kernel<MR, NR, WM, WN, false, false, OBF, true, 3, 1>), grid, dim3(64 * WM * WN), 0, st, p);
}

// column-tile width launch_conv picks for Ncol output columns
inline int conv_bn_of(int Ncol) { return Ncol <= 64 ? 64 : (Ncol <= 128 ? 128 : 256); }

template <bool OBF>
int launch_conv(ConvParams &p, bool dgrad, hipStream_t st, int ep = 0) {
    p.s_magic = magic_of(p.S);
    {
        const int taps = p.K / p.Cs, kch = rcf_kchunk(korder_chunked(p.flags), taps, p.Cs, RCF_KCHUNK_BF16);
        p.kch = kch ? kch : p.Cs;
        p.rsch = taps * p.kch;
        p.kch_magic = magic_of(p.kch);
        p.rsch_magic = magic_of(p.rsch);
    }
    if ((long)(p.K + BKT) * p.rsch >= (1L << 32)) return RCF_EINVAL;
    // 32-bit descriptor offsets: the images one row tile can touch must lie within 2 GiB of the first one
    const long per_tile_imgs = 256 / (long)p.rr + 2;
    if (per_tile_imgs * p.a_img_stride * 2 >= (1L << 31)) return RCF_EINVAL;
    const long bbytes = (long)rcf_cdiv(p.K, BKT) * p.Ncol * 64;
    if (bbytes >= (1L << 31)) return RCF_EINVAL;
    p.b_bytes = (int)bbytes;
    const bool strided = p.div > 1;
    if (ep && (strided || !OBF || p.Ncol % conv_bn_of(p.Ncol) || p.bias || p.act)) return RCF_EINVAL;
    if (p.Ncol <= 64) launch_cfg<2, 1, 2, 2, OBF>(p, strided, dgrad, st, ep);
    else if (p.Ncol <= 128) launch_cfg<2, 2, 2, 2, OBF>(p, strided, dgrad, st, ep);
    else launch_cfg<2, 4, 2, 2, OBF>(p, strided, dgrad, st, ep);
    RCF_LAUNCH_CHECK();
    return 0;
}

struct WgradPlan {
    int mr, nr, itiles, jtiles, splitk;
    long chunk;
};
WgradPlan plan_wgrad(const rcf_conv_shape *s, const rcf_conv_region *reg) {
    WgradPlan pl;
    const int ktot = s->R * s->S * s->Cin;
    pl.nr = ktot >= 256 ? 4 : (ktot >= 128 ? 2 : 1);
    // 64 output channels take the 128-row LDS-DMA kernel too when the columns allow it: these layers are bound by memory,
    // not by the half-empty tile rows (layer1 3x3 64->64: 0.104 -> 0.093 ms, 1x1 256->64: 0.089 -> 0.068)
    pl.mr = (s->Cout > 64 || (s->Cout == 64 && pl.nr >= 2)) ? 2 : 1;
    pl.itiles = rcf_cdiv(s->Cout, 64 * pl.mr);
    pl.jtiles = rcf_cdiv(ktot, 64 * pl.nr);
    const long RR = region_pixels(reg, s->Ho, s->Wo);
    const long M = (long)s->N * RR;
    const long tiles = (long)pl.itiles * pl.jtiles;
    // Split K (the pixels) over `c` workgroups per tile; 2 workgroups per CU = 512 slots.  Cost model in microseconds:
    // rounds(c) x pixels per workgroup x time per pixel (0.025 us for the 128x256 tile at the measured ~30 % of the MFMA
    // peak, proportionally less for smaller tiles down to the load-bound floor) + the fixed-order reduction, which reads
    // c copies of the weight gradient (~2 bytes/us/1e6 effective).  Small weights on many pixels (layer1) want hundreds
    // of splits, large weights on few pixels (layer4) a handful.
    const double px_us = 0.025 * fmax((double)(pl.mr * pl.nr) / 8.0, 0.35);
    const double wbytes = (double)s->Cout * ktot * 4.0;
    const long maxsk = M / 512 > 1 ? M / 512 : 1;
    const long slots = 512, hi = maxsk < 256 ? maxsk : 256;
    double best = 1e30;
    long sk = 1;
    for (long c = 1; c <= hi; ++c) {
        const double rounds = (double)((tiles * c + slots - 1) / slots);
        const double cost = rounds * (double)((M + c - 1) / c) * px_us + (c > 1 ? (double)c * wbytes / 2.0e6 + 3.0 : 0.0);
        if (cost < best - 1e-9) { best = cost; sk = c; }
    }
    long chunk = (M + sk - 1) / sk;
    chunk = (chunk + 31) / 32 * 32;
    const long img_bytes = (long)s->H * s->W * s->x_pitch * 2, dy_bytes = (long)s->Ho * s->Wo * s->y_pitch * 2;
    while (chunk > 32 && ((chunk / RR + 2) * img_bytes >= (1L << 31) || (chunk / RR + 2) * dy_bytes >= (1L << 31)))
        chunk = (chunk / 2 + 31) / 32 * 32;
    pl.splitk = (int)((M + chunk - 1) / chunk);
    pl.chunk = chunk;
    return pl;
}

bool region_ok(const rcf_conv_region *r, int H, int W) {
    return !r || (r->y0 >= 0 && r->x0 >= 0 && r->h > 0 && r->w > 0 && r->y0 + r->h <= H && r->x0 + r->w <= W &&
                  r->band >= 0 && (r->band == 0 || (2 * r->band < r->h && 2 * r->band < r->w)));
}

}  // namespace

extern "C" size_t rcf_conv_weight_bf16_bytes(int Cout, int Cin, int R, int S, int transpose) {
    if (Cout <= 0 || Cin <= 0 || R <= 0 || S <= 0) return 0;
    const long rows = transpose ? Cin : Cout, K = (long)R * S * (transpose ? Cout : Cin);
    return (size_t)((K + 31) / 32) * rows * 64;
}

namespace {
// batched weight_bf16_kernel over a table of weights (rcf_common.h: rcf_wprep_entry; out = the layout of rcf_conv_weight_bf16)
template <bool TRANSPOSE>
__global__ void __launch_bounds__(256) wprep_bf16_kernel(const rcf_wprep_entry *__restrict__ tab, int n_entries, int korder) {
    const rcf_wprep_entry t = tab[rcf_wprep_find(tab, n_entries, blockIdx.x)];
    const int rows = TRANSPOSE ? t.Cin : t.Cout;
    const int K = TRANSPOSE ? t.RS * t.Cout : t.RS * t.Cin;
    const int Cs = TRANSPOSE ? t.Cout : t.Cin, kch = rcf_kchunk(korder, t.RS, Cs, RCF_KCHUNK_BF16);
    const int KT = (K + 31) >> 5;
    const long n = (long)KT * rows * 32;
    const long step = (long)t.nblocks * 256;
    bf16_t *out = reinterpret_cast<bf16_t *>(t.out);
    for (long i = (long)(blockIdx.x - t.first_block) * 256 + threadIdx.x; i < n; i += step) {
        const int kl = (int)(i & 31);
        const long q = i >> 5;
        const int j = (int)(q % rows);
        const int kp = (int)(q / rows) * 32 + kl;
        float v = 0.f;
        if (kp < K) {
            const int k = rcf_kperm(kp, t.RS, Cs, kch);
            if (!TRANSPOSE) {
                v = t.w[(long)j * K + k];
            } else {
                const int rs = k / t.Cout, co = k - rs * t.Cout;
                v = t.w[((long)co * t.RS + rs) * t.Cin + j];
            }
        }
        out[i] = (bf16_t)v;
    }
}
}  // namespace

/* Batched rcf_conv_weight_bf16 for every weight of a model: two launches (forward operands, transposed operands).
 * tab_*: device arrays of n rcf_wprep_entry with first_block / nblocks filled per launch. */
extern "C" int rcf_conv_weights_prepare_bf16(const void *tab_fwd, int blocks_fwd, const void *tab_t, int blocks_t, int n,
                                             unsigned flags, void *stream) {
    if (!tab_fwd || !tab_t || n <= 0 || blocks_fwd <= 0 || blocks_t <= 0) return RCF_EINVAL;
    hipStream_t st = rcf_stream(stream);
    hipLaunchKernelGGL(wprep_bf16_kernel<false>, dim3((unsigned)blocks_fwd), dim3(256), 0, st, (const rcf_wprep_entry *)tab_fwd, n, (int)korder_chunked(flags));
    hipLaunchKernelGGL(wprep_bf16_kernel<true>, dim3((unsigned)blocks_t), dim3(256), 0, st, (const rcf_wprep_entry *)tab_t, n, (int)korder_chunked(flags));
    RCF_LAUNCH_CHECK();
    return 0;
}

extern "C" int rcf_conv_weight_bf16(const float *w, int Cout, int Cin, int R, int S, int transpose, void *out, unsigned flags,
                                    void *stream) {
    if (!w || !out || Cout <= 0 || Cin <= 0 || R <= 0 || S <= 0 || !rcf_aligned16(out)) return RCF_EINVAL;
    const long n = (long)rcf_conv_weight_bf16_bytes(Cout, Cin, R, S, transpose) / 2;
    const long blocks = (n + 1023) / 1024;
    const dim3 grid((unsigned)(blocks < 2048 ? blocks : 2048));
    if (transpose) hipLaunchKernelGGL(weight_bf16_kernel<true>, grid, dim3(256), 0, rcf_stream(stream), w, (bf16_t *)out, Cout, Cin, R * S, (int)korder_chunked(flags));
    else hipLaunchKernelGGL(weight_bf16_kernel<false>, grid, dim3(256), 0, rcf_stream(stream), w, (bf16_t *)out, Cout, Cin, R * S, (int)korder_chunked(flags));
    RCF_LAUNCH_CHECK();
    return 0;
}

extern "C" size_t rcf_conv2d_fwd_stats_bf16_workspace_bytes(const rcf_conv_shape *s) {
    if (check_shape(s, 4)) return 0;
    return (size_t)(rcf_cdiv((long)s->N * s->Ho * s->Wo, 128) + 64) * 2 * s->Cout * sizeof(double);
}

static int conv2d_fwd_bf16_impl(const void *x, const void *w_bf16, const float *bias, void *y, int ydt,
                               const rcf_conv_shape *s, const rcf_conv_region *region, int act, float slope, int beta,
                               double *sums, const rcf_bn_finalize *fin, void *workspace, size_t workspace_bytes,
                               void *stream) {
    if (ydt != RCF_BF16 && ydt != RCF_F32) return RCF_EINVAL;
    if (int e = check_shape(s, ydt == RCF_BF16 ? 8 : 4)) return e;
    if (!x || !w_bf16 || !y || !rcf_aligned16(x) || !rcf_aligned16(w_bf16) || !rcf_aligned16(y)) return RCF_EINVAL;
    if (ydt == RCF_BF16 ? (s->y_pitch % 8) : (s->y_pitch % 4)) return RCF_EINVAL;
    ConvParams p{};
    p.flags = s->flags;
    p.A = (const bf16_t *)x; p.Bw = (const bf16_t *)w_bf16; p.bias = bias; p.Y = y;
    p.Ncol = s->Cout; p.K = s->R * s->S * s->Cin;
    p.Ho = s->Ho; p.Wo = s->Wo; p.Hs = s->H; p.Ws = s->W; p.Cs = s->Cin; p.S = s->S;
    if (int e = set_region(p, region, s->N, s->Ho, s->Wo)) return e;
    p.up = s->stride; p.off = -s->pad; p.step = s->dil; p.div = 1;
    p.a_pitch = s->x_pitch; p.a_img_stride = (long)s->H * s->W * s->x_pitch; p.y_pitch = s->y_pitch;
    p.act = act; p.slope = slope; p.beta = beta;
    hipStream_t st = rcf_stream(stream);
    const bool stats = sums || fin;
    if (stats) {                                      // batch-norm statistics of the output from the epilogue
        if (region || bias || act || beta) return RCF_EINVAL;
        if (!workspace || workspace_bytes < rcf_conv2d_fwd_stats_bf16_workspace_bytes(s)) return RCF_EWORKSPACE;
        p.stats = (double *)workspace;
    }
    const int e = ydt == RCF_BF16 ? launch_conv<true>(p, false, st) : launch_conv<false>(p, false, st);
    if (e || !stats) return e;
    return rcf_sum_partials_bn((const double *)workspace, p.mtiles, s->Cout, sums,
                               (double *)workspace + (size_t)p.mtiles * 2 * s->Cout, fin, stream);
}

extern "C" int rcf_conv2d_fwd_bf16(const void *x, const void *w_bf16, const float *bias, void *y, int ydt,
                                   const rcf_conv_shape *s, const rcf_conv_region *region, int act, float slope, int beta,
                                   double *sums, void *workspace, size_t workspace_bytes, void *stream) {
    return conv2d_fwd_bf16_impl(x, w_bf16, bias, y, ydt, s, region, act, slope, beta, sums, nullptr, workspace,
                                workspace_bytes, stream);
}

extern "C" int rcf_conv2d_fwd_bnstats_bf16(const void *x, const void *w_bf16, void *y, int ydt, const rcf_conv_shape *s,
                                           double *sums, const rcf_bn_finalize *fin, void *workspace,
                                           size_t workspace_bytes, void *stream) {
    if (!sums && !fin) return RCF_EINVAL;
    return conv2d_fwd_bf16_impl(x, w_bf16, nullptr, y, ydt, s, nullptr, 0, 0.f, 0, sums, fin, workspace, workspace_bytes,
                                stream);
}

extern "C" size_t rcf_conv2d_dgrad_bf16_workspace_bytes(const rcf_conv_shape *s) {
    if (check_shape(s)) return 0;
    return rcf_conv_weight_bf16_bytes(s->Cout, s->Cin, s->R, s->S, 1);
}

extern "C" int rcf_conv2d_dgrad_bf16(const void *dy, const float *w, void *dx, const rcf_conv_shape *s,
                                     const rcf_conv_region *region, int beta, void *workspace, size_t workspace_bytes,
                                     void *stream) {
    if (int e = check_shape(s)) return e;
    if (!dy || !w || !dx || !rcf_aligned16(dy) || !rcf_aligned16(w) || !rcf_aligned16(dx)) return RCF_EINVAL;
    if (s->y_pitch % 8) return RCF_EINVAL;
    const void *wt = s->w_pairs_t;                        // the transposed bf16 weights, prepared once per weight update
    if (!wt) {
        const size_t need = rcf_conv2d_dgrad_bf16_workspace_bytes(s);
        if (!workspace || workspace_bytes < need || !rcf_aligned16(workspace)) return RCF_EWORKSPACE;
        if (int e = rcf_conv_weight_bf16(w, s->Cout, s->Cin, s->R, s->S, 1, workspace, s->flags, stream)) return e;
        wt = workspace;
    } else if (!rcf_aligned16(wt)) {
        return RCF_EINVAL;
    }
    ConvParams p{};
    p.flags = s->flags;
    p.A = (const bf16_t *)dy; p.Bw = (const bf16_t *)wt; p.bias = nullptr; p.Y = dx;
    p.Ncol = s->Cin; p.K = s->R * s->S * s->Cout;
    p.Ho = s->H; p.Wo = s->W; p.Hs = s->Ho; p.Ws = s->Wo; p.Cs = s->Cout; p.S = s->S;
    if (int e = set_region(p, region, s->N, s->H, s->W)) return e;
    p.up = 1; p.off = s->pad; p.step = -s->dil; p.div = s->stride;
    p.a_pitch = s->y_pitch; p.a_img_stride = (long)s->Ho * s->Wo * s->y_pitch; p.y_pitch = s->x_pitch;
    p.act = 0; p.slope = 0.f; p.beta = beta;
    return launch_conv<true>(p, true, rcf_stream(stream));
}

/* Fused forms (include/rcf_hip.h): conv -> folded batch norm -> residual -> ReLU; data gradient -> ReLU mask -> column sums */
// 64-bit words of the ReLU sign bits of an [rows][C] output in the tile order of the kernel that takes C output columns
extern "C" size_t rcf_conv_relu_bits_bytes(long rows, int C) {
    if (rows <= 0 || C <= 0) return 0;
    const int bn = conv_bn_of(C);
    return (size_t)rcf_cdiv(rows, 128) * rcf_cdiv(C, bn) * 4 * (2 * (bn / 64)) * 16 * sizeof(unsigned long long);
}

extern "C" int rcf_conv2d_fwd_affine_bf16(const void *x, const void *w_bf16, const float *scale, const float *shift,
                                          const void *residual, int res_pitch, int relu, void *y, void *relu_bits,
                                          const rcf_conv_shape *s, void *stream) {
    if (int e = check_shape(s, 8)) return e;
    if (!x || !w_bf16 || !y || !scale || !shift || !rcf_aligned16(x) || !rcf_aligned16(w_bf16) || !rcf_aligned16(y) ||
        !rcf_aligned16(scale) || !rcf_aligned16(shift)) return RCF_EINVAL;
    if (s->y_pitch % 8 || (residual && (!rcf_aligned16(residual) || res_pitch % 8 || res_pitch < s->Cout))) return RCF_EINVAL;
    if (s->stride != 1) return RCF_EINVAL;
    ConvParams p{};
    p.flags = s->flags;
    p.A = (const bf16_t *)x; p.Bw = (const bf16_t *)w_bf16; p.bias = nullptr; p.Y = y;
    p.Ncol = s->Cout; p.K = s->R * s->S * s->Cin;
    p.Ho = s->Ho; p.Wo = s->Wo; p.Hs = s->H; p.Ws = s->W; p.Cs = s->Cin; p.S = s->S;
    if (int e = set_region(p, nullptr, s->N, s->Ho, s->Wo)) return e;
    p.up = s->stride; p.off = -s->pad; p.step = s->dil; p.div = 1;
    p.a_pitch = s->x_pitch; p.a_img_stride = (long)s->H * s->W * s->x_pitch; p.y_pitch = s->y_pitch;
    p.ep_scale = scale; p.ep_shift = shift; p.ep_side = (const bf16_t *)residual; p.ep_side_pitch = res_pitch; p.ep_relu = relu;
    p.ep_bits = relu ? (unsigned long long *)relu_bits : nullptr;
    return launch_conv<true>(p, false, rcf_stream(stream), 1);
}

extern "C" size_t rcf_conv2d_dgrad_masked_bf16_workspace_bytes(const rcf_conv_shape *s) {
    if (check_shape(s)) return 0;
    return (size_t)(rcf_cdiv((long)s->N * s->H * s->W, 128) + 64) * 2 * s->Cin * sizeof(double);
}

extern "C" int rcf_conv2d_dgrad_masked_bf16(const void *dy, const void *w_t_bf16, void *dx, const rcf_conv_shape *s, int beta,
                                            const void *mask_src, int mask_pitch, const void *mask_bits, double *colsums,
                                            void *workspace, size_t workspace_bytes, void *stream) {
    if (int e = check_shape(s)) return e;
    if (!dy || !w_t_bf16 || !dx || (!mask_src && !mask_bits) || !rcf_aligned16(dy) || !rcf_aligned16(w_t_bf16) || !rcf_aligned16(dx))
        return RCF_EINVAL;
    if (mask_src && (!rcf_aligned16(mask_src) || mask_pitch % 8 || mask_pitch < s->Cin)) return RCF_EINVAL;
    if (s->y_pitch % 8 || s->stride != 1) return RCF_EINVAL;
    ConvParams p{};
    p.flags = s->flags;
    p.A = (const bf16_t *)dy; p.Bw = (const bf16_t *)w_t_bf16; p.bias = nullptr; p.Y = dx;
    p.Ncol = s->Cin; p.K = s->R * s->S * s->Cout;
    p.Ho = s->H; p.Wo = s->W; p.Hs = s->Ho; p.Ws = s->Wo; p.Cs = s->Cout; p.S = s->S;
    if (int e = set_region(p, nullptr, s->N, s->H, s->W)) return e;
    p.up = 1; p.off = s->pad; p.step = -s->dil; p.div = s->stride;
    p.a_pitch = s->y_pitch; p.a_img_stride = (long)s->Ho * s->Wo * s->y_pitch; p.y_pitch = s->x_pitch;
    p.beta = beta;
    p.ep_side = (const bf16_t *)mask_src; p.ep_side_pitch = mask_pitch;
    p.ep_bits = (unsigned long long *)mask_bits;
    if (colsums) {
        if (!workspace || workspace_bytes < rcf_conv2d_dgrad_masked_bf16_workspace_bytes(s)) return RCF_EWORKSPACE;
        p.stats = (double *)workspace;
    }
    if (int e = launch_conv<true>(p, true, rcf_stream(stream), 2)) return e;
    if (!colsums) return 0;
    // [sum | sum of squares] per input channel of what was written; the caller reads the first half
    return rcf_sum_partials_bn((const double *)workspace, p.mtiles, s->Cin, colsums,
                               (double *)workspace + (size_t)p.mtiles * 2 * s->Cin, nullptr, stream);
}

extern "C" size_t rcf_conv2d_wgrad_bf16_workspace_bytes(const rcf_conv_shape *s, const rcf_conv_region *region) {
    if (check_shape(s) || !region_ok(region, s->Ho, s->Wo)) return 0;
    const WgradPlan pl = plan_wgrad(s, region);
    if (pl.splitk <= 1) return 0;
    return (size_t)pl.splitk * s->Cout * s->R * s->S * s->Cin * sizeof(float);
}

extern "C" int rcf_conv2d_wgrad_bf16(const void *x, const void *dy, float *dw, const rcf_conv_shape *s,
                                     const rcf_conv_region *region, int beta, void *workspace, size_t workspace_bytes,
                                     void *stream) {
    if (int e = check_shape(s)) return e;
    if (!x || !dy || !dw || !rcf_aligned16(x) || !rcf_aligned16(dy) || !rcf_aligned16(dw)) return RCF_EINVAL;
    if (s->y_pitch % 8 || !region_ok(region, s->Ho, s->Wo)) return RCF_EINVAL;
    const WgradPlan pl = plan_wgrad(s, region);
    const size_t need = rcf_conv2d_wgrad_bf16_workspace_bytes(s, region);
    if (need > 0 && (!workspace || workspace_bytes < need || !rcf_aligned16(workspace))) return RCF_EWORKSPACE;
    hipStream_t st = rcf_stream(stream);
    WgradParams p{};
    p.X = (const bf16_t *)x; p.DY = (const bf16_t *)dy;
    p.OUT = pl.splitk > 1 ? (float *)workspace : dw;
    p.Cout = s->Cout; p.Cin = s->Cin; p.R = s->R; p.S = s->S;
    p.H = s->H; p.W = s->W; p.Ho = s->Ho; p.Wo = s->Wo; p.stride = s->stride; p.pad = s->pad; p.dil = s->dil;
    p.x_pitch = s->x_pitch; p.dy_pitch = s->y_pitch;
    p.ry0 = region ? region->y0 : 0; p.rx0 = region ? region->x0 : 0;
    p.rh = region ? region->h : s->Ho; p.rw = region ? region->w : s->Wo;
    p.rband = region ? region->band : 0; p.rr = region_pixels(region, s->Ho, s->Wo);
    p.M = (long)s->N * p.rr; p.chunk = pl.chunk; p.itiles = pl.itiles; p.jtiles = pl.jtiles;
    p.split_stride = (long)s->Cout * s->R * s->S * s->Cin; p.beta = beta;
    p.Ktot = s->R * s->S * s->Cin;
    p.sched = 1;
    p.xcd_map = (s->flags & RCF_CONV_NO_WGRAD_XCD) ? 0 : 1;
    const dim3 grid((unsigned)(pl.itiles * pl.jtiles), 1u, (unsigned)pl.splitk);
#define RCF_WG(MRv, NRv)                                                                              \
    do {                                                                                              \
        if (region) hipLaunchKernelGGL((wgrad_bf16_kernel<MRv, NRv, true>), grid, dim3(256), 0, st, p);  \
        else hipLaunchKernelGGL((wgrad_bf16_kernel<MRv, NRv, false>), grid, dim3(256), 0, st, p);        \
    } while (0)
    const bool onetap = s->Cin % (64 * pl.nr) == 0;
    p.cblocks = onetap && s->R * s->S > 1 && p.xcd_map ? s->Cin / (64 * pl.nr) : 0;
#define RCF_WGD(NRv)                                                                                              \
    do {                                                                                                          \
        if (region && onetap) hipLaunchKernelGGL((wgrad_bf16_dma_kernel<NRv, true, true>), grid, dim3(256), 0, st, p);   \
        else if (region) hipLaunchKernelGGL((wgrad_bf16_dma_kernel<NRv, true, false>), grid, dim3(256), 0, st, p);       \
        else if (onetap) hipLaunchKernelGGL((wgrad_bf16_dma_kernel<NRv, false, true>), grid, dim3(256), 0, st, p);       \
        else hipLaunchKernelGGL((wgrad_bf16_dma_kernel<NRv, false, false>), grid, dim3(256), 0, st, p);                  \
    } while (0)
    if (pl.mr == 2 && pl.nr == 4) RCF_WGD(4);
    else if (pl.mr == 2 && pl.nr == 2) RCF_WGD(2);
    else if (pl.mr == 2) RCF_WG(2, 1);
    else if (pl.nr == 4) RCF_WG(1, 4);
    else if (pl.nr == 2) RCF_WG(1, 2);
    else RCF_WG(1, 1);
#undef RCF_WG
#undef RCF_WGD
    RCF_LAUNCH_CHECK();
    if (pl.splitk > 1) {
        const long n4 = p.split_stride / 4;
        const int bt = n4 < (1 << 17) ? 64 : 256;
        const int blocks = (int)((n4 + bt - 1) / bt < 4096 ? (n4 + bt - 1) / bt : 4096);
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(bt), 0, st, (const float *)workspace, dw, n4,
                           p.split_stride, pl.splitk, beta);
        RCF_LAUNCH_CHECK();
    }
    return 0;
}
