// Shared device/host helpers for librcf_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/rcf_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define RCF_LAUNCH_CHECK()                         \
    do {                                           \
        hipError_t e__ = hipGetLastError();        \
        if (e__ != hipSuccess) return (int)e__;    \
    } while (0)

static inline hipStream_t rcf_stream(void *s) { return (hipStream_t)s; }
// bn.hip: partial [chunks][2C] (per conv row tile: sum | sum of squares) -> sums[2C] (may be NULL with fin) and, with fin,
// the batch-norm constants + running statistics + num_batches_tracked, in one launch.  scratch: 64 rows of 2C doubles.
int rcf_sum_partials_bn(const double *partial, int chunks, int C, double *sums, double *scratch,
                        const rcf_bn_finalize *fin, void *stream);

// csrc/crf_sort.hip: rocPRIM radix sort / inclusive scan for the sort-based lattice build (csrc/crf.hip)
size_t rcf_crf_sort_tmp_bytes(size_t n);
int rcf_crf_sort_pairs_u64(void *tmp, size_t tmp_bytes, const unsigned long long *k_in, unsigned long long *k_out,
                           const unsigned *v_in, unsigned *v_out, size_t n, int end_bit, hipStream_t st);
int rcf_crf_inclusive_scan_i32(void *tmp, size_t tmp_bytes, const int *in, int *out, size_t n, hipStream_t st);

// csrc/thin.hip: streaming kernels for 1x1 convs with 4 / 8 / 16 output channels (the decode heads' classifiers); fp32 tensors.
// The fp32 conv entry points hand over when rcf_thin_ok(shape) holds, the launch covers the whole tensor and nothing is fused.
bool rcf_thin_ok(const rcf_conv_shape *s);
size_t rcf_thin_wgrad_workspace_bytes(const rcf_conv_shape *s);
int rcf_thin_fwd(const float *x, const float *w, const float *bias, float *y, const rcf_conv_shape *s, int beta, hipStream_t st);
int rcf_thin_dgrad(const float *dy, const float *w, float *dx, const rcf_conv_shape *s, int beta, hipStream_t st);
int rcf_thin_wgrad(const float *x, const float *dy, float *dw, const rcf_conv_shape *s, int beta, void *workspace, size_t workspace_bytes,
                   hipStream_t st);

static inline int rcf_cdiv(long a, long b) { return (int)((a + b - 1) / b); }
static inline bool rcf_aligned16(const void *p) { return (((uintptr_t)p) & 15) == 0; }

// 64-lane wavefront reductions (DPP/shuffle based)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---- mixed-precision storage (the bf16 training step: BASELINE configs[2]) --------------------------------------------
// Activations / activation gradients may be stored as bf16 (dtype code RCF_BF16) instead of fp32 (RCF_F32); every kernel
// computes in fp32.  ld4 / st4 move FOUR consecutive channels (16 B of fp32, 8 B of bf16); bf16 -> fp32 is exact,
// fp32 -> bf16 rounds to nearest even (v_cvt_pk_bf16_f32).
// The 16-bit storage type is a BUILD parameter (round 6): librcf_hip.so stores bf16 (the default: fp32's exponent range, no loss
// scaling), librcf_hip_f16.so -- the four sources that touch 16-bit tensors (bn, spatial, igemm_bf16, foldbn) compiled again with
// -DRCF_HALF_F16 -- stores IEEE fp16: Lightning's `precision: 16` of the STv2 / FBMS configs (fp16 autocast + GradScaler,
// configs/rcf_stv2/rcf_stage1.yaml:57-60).  The type keeps its historical name `bf16_t` and the dtype code RCF_BF16 means "the
// 16-bit type of the library the call goes to"; the MFMA instruction (v_mfma_f32_32x32x16_{bf16,f16}) and the widening
// conversions (a shift for bf16, v_cvt for fp16) are the only places that differ.
#ifdef RCF_HALF_F16
typedef _Float16 bf16_t;
typedef _Float16 bf16x4_t __attribute__((ext_vector_type(4)));
#define RCF_MFMA_32X32X16_H(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
#else
typedef __bf16 bf16_t;
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
#define RCF_MFMA_32X32X16_H(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
#endif
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
typedef bf16_t h16x2_t __attribute__((ext_vector_type(2)));
// four packed 16-bit values (two dwords) -> fp32 (exact in both types)
__device__ __forceinline__ f32x4 rcf_widen4(unsigned a, unsigned b) {
#ifdef RCF_HALF_F16
    const h16x2_t ha = __builtin_bit_cast(h16x2_t, a), hb = __builtin_bit_cast(h16x2_t, b);
    return f32x4{(float)ha[0], (float)ha[1], (float)hb[0], (float)hb[1]};
#else
    return f32x4{__uint_as_float(a << 16), __uint_as_float(a & 0xffff0000u), __uint_as_float(b << 16), __uint_as_float(b & 0xffff0000u)};
#endif
}

template <typename T> __device__ __forceinline__ f32x4 ld4(const T *p);
template <> __device__ __forceinline__ f32x4 ld4<float>(const float *p) { return *reinterpret_cast<const f32x4 *>(p); }
template <> __device__ __forceinline__ f32x4 ld4<bf16_t>(const bf16_t *p) {
    const u32x2_t r = *reinterpret_cast<const u32x2_t *>(p);
    return rcf_widen4(r[0], r[1]);
}
template <typename T> __device__ __forceinline__ void st4(T *p, f32x4 v);
template <> __device__ __forceinline__ void st4<float>(float *p, f32x4 v) { *reinterpret_cast<f32x4 *>(p) = v; }
template <> __device__ __forceinline__ void st4<bf16_t>(bf16_t *p, f32x4 v) {
    *reinterpret_cast<bf16x4_t *>(p) = __builtin_convertvector(v, bf16x4_t);
}
template <typename T> __device__ __forceinline__ float ld1(const T *p) { return (float)*p; }
template <typename T> __device__ __forceinline__ void st1(T *p, float v) { *p = (T)v; }

// V consecutive channels (V = 4 or 8) as fp32: one 16-byte access moves 8 bf16 channels
template <int V> struct fvec { f32x4 q[V / 4]; };
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef bf16_t bf16x8_t __attribute__((ext_vector_type(8)));
template <typename T, int V> __device__ __forceinline__ fvec<V> ldv(const T *p) {
    fvec<V> r;
    if constexpr (V == 8 && sizeof(T) == 2) {
        const u32x4_t w = *reinterpret_cast<const u32x4_t *>(p);
        r.q[0] = rcf_widen4(w[0], w[1]);
        r.q[1] = rcf_widen4(w[2], w[3]);
    } else {
#pragma unroll
        for (int h = 0; h < V / 4; ++h) r.q[h] = ld4(p + 4 * h);
    }
    return r;
}
template <typename T, int V> __device__ __forceinline__ void stv(T *p, const fvec<V> &v) {
    if constexpr (V == 8 && sizeof(T) == 2) {
        const bf16x4_t a = __builtin_convertvector(v.q[0], bf16x4_t), b = __builtin_convertvector(v.q[1], bf16x4_t);
        const u32x2_t ua = __builtin_bit_cast(u32x2_t, a), ub = __builtin_bit_cast(u32x2_t, b);
        *reinterpret_cast<u32x4_t *>(p) = u32x4_t{ua[0], ua[1], ub[0], ub[1]};
    } else {
#pragma unroll
        for (int h = 0; h < V / 4; ++h) st4(p + 4 * h, v.q[h]);
    }
}

// dtype-code dispatch for the *_mp entry points: expands `CALL(T)` with T = float / bf16_t
#define RCF_DISPATCH1(dt, CALL)                          \
    do {                                                 \
        if ((dt) == RCF_F32) { CALL(float); }            \
        else if ((dt) == RCF_BF16) { CALL(bf16_t); }     \
        else return RCF_EINVAL;                          \
    } while (0)
// (x side, y side): (f32, f32), (bf16, bf16) and (f32, bf16) -- an fp32 conv output normalised into bf16 activations
#define RCF_DISPATCH2(xdt, ydt, CALL)                                              \
    do {                                                                           \
        if ((xdt) == RCF_F32 && (ydt) == RCF_F32) { CALL(float, float); }          \
        else if ((xdt) == RCF_BF16 && (ydt) == RCF_BF16) { CALL(bf16_t, bf16_t); } \
        else if ((xdt) == RCF_F32 && (ydt) == RCF_BF16) { CALL(float, bf16_t); }   \
        else return RCF_EINVAL;                                                    \
    } while (0)

// ---- batched weight preparation (rcf_conv_weights_prepare_*): one launch per derived layout for ALL conv weights of a model
// instead of one per layer -- the derived operands (ranges, fp16 pair planes, bf16 copies) are rebuilt after every optimizer
// step, and ~60 layers x 4 five-microsecond launches were 1-2 % of a training step.  A device table holds one entry per
// weight; block b of the launch serves entry e with first_block[e] <= b < first_block[e] + nblocks[e].
struct rcf_wprep_entry {
    const float *w;            // [Cout][R][S][Cin] fp32 master weight
    void *out;                 // the layout this table is for (see rcf_conv_weights_prepare_* in include/rcf_hip.h)
    unsigned *amax;            // the weight's range: raw bits of max |w| (written by the range launch, read by the pair launches)
    int Cout, Cin, RS;
    int first_block, nblocks;
    int flags;                 // bit 0: also write the plane-separated (pairs2) half of `out`
    int pad_[2];
};
static_assert(sizeof(rcf_wprep_entry) == 56 || sizeof(rcf_wprep_entry) == 64, "mirrored by ctypes in _lib.py");

// K order of the implicit-GEMM forward / data-gradient convs.  Natural: k = tap * Cs + c (the weight's memory order) -- a
// workgroup streams ALL channels of its pixels for tap 0, then for tap 1, ...: by the time the next tap asks for the same
// pixels (shifted by the dilation) they have left the XCD's 4 MB L2, and a 3x3 conv over 2048 / 4096 channels pulls its input
// through the fabric 8 - 9 times (profiles/r03_pmc_traffic_by_layer_before.txt).  Chunked: K runs over (channel chunk of
// 32 / 64, tap, channel inside the chunk) -- the nine taps of one chunk follow each other, the row tiles running together on an
// XCD are neighbours (mtiles8), so a tap's pixels are still in L2 from the previous tap or from the neighbouring tile.
// rcf_kchunk: chunk width of a (taps, channels-per-tap) pair, 0 = natural order; the weight preparation kernels and the conv
// kernels evaluate the same rule.  rcf_kperm: position in the K loop -> natural index tap * Cs + c.
// Chunk width: 128 bytes of a pixel (one cache line) in fp32, 128 bytes = two K-steps in bf16 -- the narrower the chunk, the
// more workgroups of an XCD can drift apart by a chunk before their pixels fall out of L2 (fp32 at 64 channels: a 3x3
// 4096 -> 256 conv still fetched 7 x its input, profiles/r03_pmc_traffic_by_layer.txt).
constexpr int RCF_KCHUNK_F32 = 32, RCF_KCHUNK_BF16 = 64;
__host__ __device__ inline int rcf_kchunk(int mode, int RS, int Cs, int width) {
    // from 256 channels per tap up: below that a tap's pixels survive in L2 anyway (fabric traffic 1.1 - 1.8 x the
    // algorithmic bytes either way), and the narrow convs keep the summation order their golden vectors were checked under
    // (the flow head's 64 -> 64 convs feed a LeakyReLU whose branch at |x| ~ 1e-7 decides 1e-4 of a bias gradient)
    return (mode && RS > 1 && Cs >= 256 && Cs % width == 0) ? width : 0;
}
// rcf_kpos: natural index -> position in the K loop (the inverse of rcf_kperm)
__host__ __device__ inline int rcf_kpos(int k, int RS, int Cs, int kch) {
    if (!kch) return k;
    const int tap = k / Cs, c = k - tap * Cs;
    const int q = c / kch;
    return q * (RS * kch) + tap * kch + (c - q * kch);
}
__host__ __device__ inline int rcf_kperm(int kp, int RS, int Cs, int kch) {
    if (!kch) return kp;
    const int per = RS * kch;
    const int q = kp / per, rem = kp - q * per;
    const int tap = rem / kch;
    return tap * Cs + q * kch + (rem - tap * kch);
}

// Forward / data-gradient grids: which XCD computes which (row tile, column tile).  Default (colmap 0): XCD x walks the row
// tiles of ITS band [x mtiles8, (x + 1) mtiles8), all column tiles of a row tile one after the other -- the activation tile
// stays in that XCD's L2 across the column tiles, the WEIGHTS stream through every XCD once per row tile.  That is right
// while the weights fit L2 next to the activations (4 MB), and wrong for the data gradient of a conv with thousands of input
// channels: 16 column tiles of 2304 x 256 weights = 19 - 38 MB per row tile, 400 - 800 row tiles -> 5 - 9 x the launch's
// bytes through the fabric (profiles/r03_pmc_traffic_by_layer.txt).  colmap 1: XCD x owns the column tiles
// [x ntiles/8, (x + 1) ntiles/8) of EVERY row tile -- its share of the weights stays in its L2, the (small) activation
// operand is fetched by all eight XCDs.  rcf_colmap_pays: the byte model that picks it.
__device__ __forceinline__ void rcf_conv_tile(int bid, int mtiles8, int ntiles, int colmap, int &tile_m, int &tile_n) {
    if (colmap) {
        const int x = bid & 7, k = bid >> 3, nt8 = ntiles >> 3;
        tile_m = k / nt8;
        tile_n = x * nt8 + (k - tile_m * nt8);
    } else {
        const int grp = bid / (8 * ntiles), rem = bid - grp * 8 * ntiles;
        tile_n = rem >> 3;
        tile_m = (rem & 7) * mtiles8 + grp;
    }
}
// a_bytes / b_bytes: the activation operand (rows x channels per tap) and the whole weight operand as the kernel reads them
static inline int rcf_colmap_pays(int mode, long a_bytes, long b_bytes, int mtiles, int ntiles) {
    if (!mode || ntiles < 8 || ntiles % 8) return 0;
    if (b_bytes < (8L << 20)) return 0;                 // the weights (nearly) fit L2: row bands already re-use them
    return 8 * a_bytes < b_bytes * (mtiles / 4);        // a quarter of the row-band form's weight re-reads still hit L2
}

// Weight-gradient grids are (tiles, 1, splits): a split is a range of pixels of dy and x, every tile of a split reads that
// range.  Workgroups go round-robin over the 8 XCDs in dispatch order (L = x + tiles * z -> XCD L % 8), so with the plain
// mapping every XCD's L2 fetches every split's pixels.  mode 1: XCD x takes a CONTIGUOUS run of the (split, tile) sequence
// -- the workgroups running together on one XCD walk the same pixels, which then cross the fabric once per XCD that needs
// them instead of once per XCD (counters: profiles/r03_pmc_wgrad_xcd.txt).  A bijection of [0, tiles * splits) for any count.
__device__ __forceinline__ void rcf_wgrad_item(int mode, int &tile, int &split) {
    const unsigned tiles = gridDim.x, total = tiles * gridDim.z;
    const unsigned L = blockIdx.x + tiles * blockIdx.z;
    if (mode == 0 || total < 16) {
        tile = (int)blockIdx.x;
        split = (int)blockIdx.z;
        return;
    }
    const unsigned q = total >> 3, r = total & 7u, x = L & 7u;
    const unsigned seq = x * q + (x < r ? x : r) + (L >> 3);
    const unsigned sp = seq / tiles;
    split = (int)sp;
    tile = (int)(seq - sp * tiles);
}

// Order of the output tiles inside a split.  With the (tap, channel) pairs as GEMM columns, column tile tj = tap * cblocks + cb
// reads channel block cb of x shifted by the tap's offset: the nine tiles of ONE channel block read the same pixels of x.
// cblocks > 0 (every column tile lies in one tap, several taps): tiles are numbered (channel block, tap, row tile), so the
// workgroups running together on an XCD cover few channel blocks under all their taps -- a pixel of x crosses the fabric once
// per XCD instead of once per tap (a 3x3 4096 -> 256 weight gradient: 10.5 x its algorithmic bytes before,
// profiles/r03_pmc_traffic_by_layer_before.txt).  cblocks == 0: row-tile-major, as the grid was laid out originally.
__device__ __forceinline__ void rcf_wgrad_tile_ij(int tile, int itiles, int jtiles, int cblocks, int &ti, int &tj) {
    if (cblocks > 0) {
        const int taps = jtiles / cblocks, per_cb = taps * itiles;
        const int cb = tile / per_cb, rem = tile - cb * per_cb;
        const int tap = rem / itiles;
        ti = rem - tap * itiles;
        tj = tap * cblocks + cb;
    } else {
        ti = tile / jtiles;
        tj = tile - ti * jtiles;
    }
}

__device__ __forceinline__ int rcf_wprep_find(const rcf_wprep_entry *__restrict__ tab, int n, int block) {
    int lo = 0, hi = n - 1;
    while (lo < hi) {                                      // last entry with first_block <= block
        const int mid = (lo + hi + 1) >> 1;
        if (tab[mid].first_block <= block) lo = mid;
        else hi = mid - 1;
    }
    return lo;
}
