// Training-mode batch norm on NHWC fp32, with the ReLU, the residual add and the Dropout2d channel
// scale of the reference fused in.  Reference: (Sync)BatchNorm built at models/resnet.py:159-162 and
// by mmcv ConvModule (models/fcn_head.py:107-130), ReLU + `out += identity` models/resnet.py:268-300,
// nn.Dropout2d models/decode_head.py:84-85.
//
// All kernels are HBM-bound column reductions / element-wise passes over a [rows][C] matrix whose
// rows are pixels: a wavefront reads 64 x 16 B along C (coalesced), statistics accumulate in fp64
// per thread, are combined through LDS per block and written as per-block partials that a second
// tiny kernel sums in a fixed order (deterministic, no atomics).  The fp64 sums are what a
// data-parallel run all-reduces (SyncBN).
#include "rcf_common.h"

namespace {

constexpr int RED_THREADS = 256;

// Order in which a streaming kernel walks the rows of a [rows][C] tensor.  The 256 MB Infinity Cache sits on the memory side
// and keeps what the previous kernel wrote or read last (tools/mall_probe.py: a 105 MB tensor read right after it was written
// streams at 6.2 TB/s, 3.4 after unrelated traffic); a tensor larger than the cache that is produced front to back and then
// consumed front to back gets nothing out of it -- its head has been evicted by its tail.  So consumers walk BACKWARDS
// through what the producer just wrote.  The conv kernels produce eight bands at once (XCD x walks the row tiles
// [x mtiles8, (x + 1) mtiles8) upwards), hence:
//   mode 0: position l = row l.
//   mode 1 / 2: groups of `group` rows, dealt round-robin to eight bands of `band` rows (= mtiles8 * 128, the convs' bands);
//               inside its band a group's place rises (1) or falls (2) with time.
struct Sweep {
    int mode, group;
    long band;          // rows per band, a multiple of group
    __device__ __forceinline__ long positions(long rows) const { return mode ? 8 * band : rows; }
    __device__ __forceinline__ long row(long l) const {
        if (!mode) return l;
        const long g = l / group, w = l - g * group;
        const long b = g & 7, pos = g >> 3, per = band / group;
        return b * band + (mode == 2 ? per - 1 - pos : pos) * group + w;
    }
};
// The banded orders (forward apply falling, backward reduce falling, backward apply rising) apply to tensors of 192 MB and more;
// the call's flags can switch them off (RCF_BN_SWEEP_OFF) or on for every tensor of 8192 rows and more (RCF_BN_SWEEP_ALWAYS: tests).
// Tensors the cache holds whole are served from it in any order, and the banded walk costs them 3 - 5 % (measured per launch,
// profiles/r03_bn_sweep.txt: launches under 80 us lose, those over 100 us gain 3 - 15 %): plain order below 192 MB.
Sweep make_sweep(int mode, long rows, int group, long row_bytes, unsigned flags) {
    Sweep s{0, 1, 0};
    if ((flags & RCF_BN_SWEEP_OFF) || mode == 0 || group <= 0 || 128 % group || rows < 8 * 1024 ||
        (!(flags & RCF_BN_SWEEP_ALWAYS) && rows * row_bytes < (192L << 20)))
        return s;
    const long mtiles = (rows + 127) / 128, mtiles8 = (mtiles + 7) / 8;
    s.mode = mode;
    s.group = group;
    s.band = mtiles8 * 128;
    return s;
}

struct ColGeom {
    int cvB;      // channel vectors (V channels each) handled per block (<= 64)
    int RG;       // row groups per block = 256 / cvB
    int cgroups;  // blocks along C
    int chunks;   // blocks along rows
    long rows_per_chunk;
};

ColGeom col_geom(long rows, int C, int V = 4) {
    ColGeom g;
    const int CV = C / V;
    g.cvB = CV < 64 ? CV : 64;
    g.RG = RED_THREADS / g.cvB;
    g.cgroups = (CV + g.cvB - 1) / g.cvB;
    long chunks = 1024 / g.cgroups;
    const long min_rows = (long)g.RG * 8;
    if (chunks > (rows + min_rows - 1) / min_rows) chunks = (rows + min_rows - 1) / min_rows;
    if (chunks < 1) chunks = 1;
    g.rows_per_chunk = (rows + chunks - 1) / chunks;
    g.chunks = (int)((rows + g.rows_per_chunk - 1) / g.rows_per_chunk);
    return g;
}

// Generic two-value column reduction.  F(row, c, out a[V], out b[V]) produces the two addends of V consecutive channels.
template <class F, int V>
__global__ void __launch_bounds__(RED_THREADS) colreduce2_kernel(F f, long rows, int C, int cvB, int RG,
                                                                 long rows_per_chunk, double *__restrict__ partial,
                                                                 Sweep sw = Sweep{0, 1, 0}) {
    __shared__ double red[RED_THREADS * 2 * V];
    const int tid = threadIdx.x;
    const int cv = tid % cvB, rg = tid / cvB;
    const int c0 = (blockIdx.y * cvB + cv) * V;
    const long r0 = (long)blockIdx.x * rows_per_chunk;
    const long r1 = min(rows, r0 + rows_per_chunk);
    double sa[V], sb[V];
#pragma unroll
    for (int e = 0; e < V; ++e) { sa[e] = 0; sb[e] = 0; }
    if (rg < RG && c0 < C) {
        f.init(c0);                      // per-channel constants into registers, once per thread
        if (sw.mode) {
            // all blocks advance through the sweep together (block b: groups b, b + chunks, ..), not each through its own chunk
            const long lend = sw.positions(rows), lstep = (long)gridDim.x * RG;
            for (long l = (long)blockIdx.x * RG + rg; l < lend; l += lstep) {
                const long r = sw.row(l);
                if (r >= rows) continue;
                float a[V], b[V];
                f(r, c0, a, b);
#pragma unroll
                for (int e = 0; e < V; ++e) {
                    sa[e] += (double)a[e];
                    sb[e] += (double)b[e];
                }
            }
        } else {
        for (long r = r0 + rg; r < r1; r += RG) {
            float a[V], b[V];
            f(r, c0, a, b);
#pragma unroll
            for (int e = 0; e < V; ++e) {
                sa[e] += (double)a[e];
                sb[e] += (double)b[e];
            }
        }
        }
    }
#pragma unroll
    for (int e = 0; e < V; ++e) {
        red[tid * 2 * V + e] = sa[e];
        red[tid * 2 * V + V + e] = sb[e];
    }
    __syncthreads();
    if (rg == 0 && c0 < C) {
        for (int g = 1; g < RG; ++g) {
#pragma unroll
            for (int e = 0; e < V; ++e) {
                sa[e] += red[(g * cvB + cv) * 2 * V + e];
                sb[e] += red[(g * cvB + cv) * 2 * V + V + e];
            }
        }
        double *dst = partial + (long)blockIdx.x * 2 * C;
#pragma unroll
        for (int e = 0; e < V; ++e) {
            dst[c0 + e] = sa[e];
            dst[C + c0 + e] = sb[e];
        }
    }
}

// out[i] = sum_k partial[k][i]: COLS outputs x (256 / COLS) chunk-slices per block, fixed summation order.  The time of this
// kernel is its DEPENDENT load rounds (chunks / slices / 4 chains), not its bytes: 32 columns x 8 slices over the 1024 partial
// rows a batch-norm backward reduction leaves took 12.5 us for 4 MB (round 5 step profile: 58 launches, 0.72 ms); launch_partial_sum
// narrows the column group until enough workgroups run and a thread has at most ~32 rows (round 6).
template <int COLS>
__global__ void __launch_bounds__(256) partial_sum_kernel(const double *__restrict__ partial, int chunks, int n,
                                                          double *__restrict__ out) {
    constexpr int SL = 256 / COLS;
    __shared__ double sh[256];
    const int j = threadIdx.x % COLS, s = threadIdx.x / COLS;
    const int i = blockIdx.x * COLS + j;
    double acc = 0;
    if (i < n) {
        // four independent chains (loads in flight), combined in a fixed order
        double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
        int k = s;
        for (; k + 3 * SL < chunks; k += 4 * SL) {
            a0 += partial[(long)k * n + i];
            a1 += partial[(long)(k + SL) * n + i];
            a2 += partial[(long)(k + 2 * SL) * n + i];
            a3 += partial[(long)(k + 3 * SL) * n + i];
        }
        for (; k < chunks; k += SL) a0 += partial[(long)k * n + i];
        acc = (a0 + a1) + (a2 + a3);
    }
    sh[threadIdx.x] = acc;
    __syncthreads();
    if (s == 0 && i < n) {
        double t = 0;
#pragma unroll
        for (int q = 0; q < SL; ++q) t += sh[q * COLS + j];
        out[i] = t;
    }
}
// the column group of a reduction over `chunks` rows of n values: 32 (one 256-byte row piece per block and row) unless that
// leaves a thread more than 32 rows or the chip fewer than 64 workgroups -- then 16, then 8 columns (64-byte pieces)
inline void launch_partial_sum(const double *partial, int chunks, int n, double *out, hipStream_t st) {
    int cols = 32;
    while (cols > 8 && (chunks / (256 / cols) > 32 || rcf_cdiv(n, cols) < 64)) cols >>= 1;
    if (cols == 32) hipLaunchKernelGGL(partial_sum_kernel<32>, dim3(rcf_cdiv(n, 32)), dim3(256), 0, st, partial, chunks, n, out);
    else if (cols == 16) hipLaunchKernelGGL(partial_sum_kernel<16>, dim3(rcf_cdiv(n, 16)), dim3(256), 0, st, partial, chunks, n, out);
    else hipLaunchKernelGGL(partial_sum_kernel<8>, dim3(rcf_cdiv(n, 8)), dim3(256), 0, st, partial, chunks, n, out);
}

// out[g][i] = sum of partial[k][i] over the g-th group of `per` rows (blockIdx.y = g); same fixed order as above
__global__ void __launch_bounds__(256) partial_sum_groups_kernel(const double *__restrict__ partial, int chunks, int per,
                                                                 int n, double *__restrict__ out) {
    __shared__ double sh[256];
    const int j = threadIdx.x & 31, s = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + j;
    const int k0 = blockIdx.y * per, k1 = min(chunks, k0 + per);
    double acc = 0;
    if (i < n) {
        double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
        int k = k0 + s;
        for (; k + 24 < k1; k += 32) {
            a0 += partial[(long)k * n + i];
            a1 += partial[(long)(k + 8) * n + i];
            a2 += partial[(long)(k + 16) * n + i];
            a3 += partial[(long)(k + 24) * n + i];
        }
        for (; k < k1; k += 8) a0 += partial[(long)k * n + i];
        acc = (a0 + a1) + (a2 + a3);
    }
    sh[threadIdx.x] = acc;
    __syncthreads();
    if (s == 0 && i < n) {
        double t = 0;
#pragma unroll
        for (int q = 0; q < 8; ++q) t += sh[q * 32 + j];
        out[(long)blockIdx.y * n + i] = t;
    }
}

// block-wide max of non-negative raw float bits -> one atomicMax (the operand range of the fp16-pair conv kernels
// that read this tensor next; include/rcf_hip.h rcf_conv_shape).  Every thread of the block must call it.
__device__ __forceinline__ void block_amax(unsigned mx, unsigned *__restrict__ amax) {
    __shared__ unsigned sh_amax[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = max(mx, (unsigned)__shfl_xor((int)mx, o));
    if ((threadIdx.x & 63) == 0) sh_amax[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        // only while it raises the running maximum: thousands of atomics onto one address are served one after the other
        const unsigned m = max(max(sh_amax[0], sh_amax[1]), max(sh_amax[2], sh_amax[3]));
        if (m > __hip_atomic_load(amax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(amax, m);
    }
}

template <typename XT, int V>
struct StatsOp {
    const XT *x;
    int pitch;
    __device__ void init(int) {}
    __device__ void operator()(long r, int c0, float (&a)[V], float (&b)[V]) const {
        const fvec<V> v = ldv<XT, V>(x + r * pitch + c0);
#pragma unroll
        for (int e = 0; e < V; ++e) {
            a[e] = v.q[e >> 2][e & 3];
            b[e] = a[e] * a[e];
        }
    }
};

// XT: storage type of the conv-output side (x, dx); YT: of the activation side (y, dy, residual, dres)
template <typename XT, typename YT, int V>
struct BwdOp {
    const YT *dy;
    const XT *x;
    const YT *y;
    const float *mean, *invstd, *scale;
    int dy_pitch, x_pitch, y_pitch, relu, C;
    long rows_per_image;
    const unsigned char *mask;     // [rows][C/4]: bit e = output 4j+e was positive (replaces the read of y)
    fvec<V> mu, is;                // this thread's channels (init)
    __device__ void init(int c0) {
        mu = ldv<float, V>(mean + c0);
        is = ldv<float, V>(invstd + c0);
    }
    __device__ void operator()(long r, int c0, float (&a)[V], float (&b)[V]) const {
        fvec<V> g = ldv<YT, V>(dy + r * dy_pitch + c0);
        const fvec<V> xv = ldv<XT, V>(x + r * x_pitch + c0);
        if (scale) {
            const fvec<V> sc = ldv<float, V>(scale + (r / rows_per_image) * C + c0);
#pragma unroll
            for (int h = 0; h < V / 4; ++h) g.q[h] *= sc.q[h];
        }
        if (relu && mask) {
            const unsigned char *mp = mask + r * (C >> 2) + (c0 >> 2);
#pragma unroll
            for (int h = 0; h < V / 4; ++h) {
                const unsigned m = mp[h];
#pragma unroll
                for (int e = 0; e < 4; ++e) g.q[h][e] = (m >> e) & 1u ? g.q[h][e] : 0.f;
            }
        } else if (relu) {
            const fvec<V> yv = ldv<YT, V>(y + r * y_pitch + c0);
#pragma unroll
            for (int e = 0; e < V; ++e) g.q[e >> 2][e & 3] = yv.q[e >> 2][e & 3] > 0.f ? g.q[e >> 2][e & 3] : 0.f;
        }
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const int h = e >> 2, k = e & 3;
            a[e] = g.q[h][k];
            b[e] = g.q[h][k] * ((xv.q[h][k] - mu.q[h][k]) * is.q[h][k]);
        }
    }
};

// ---- the join of a stage's first block and its (lazy) downsample norm in ONE reduction: both norms' backward sums are taken of
// the same g = dy under the join's sign bits, against their own inputs x (conv3's output) and x2 (the downsample conv's).
// Four column sums [sum g | sum g xhat | sum g | sum g xhat2] (the first repeated: each norm's [2C] pair is contiguous), every one
// accumulated from the same addends in the same order as rcf_bn_bwd_reduce_mp would: bit-identical to the two passes.
template <typename XT, typename YT, int V>
struct BwdOp2 {
    const YT *dy;
    const XT *x, *x2;
    const float *mean, *invstd, *mean2, *invstd2;
    int dy_pitch, x_pitch, x2_pitch, C;
    const unsigned char *mask;
    fvec<V> mu, is, mu2, is2;
    __device__ void init(int c0) {
        mu = ldv<float, V>(mean + c0); is = ldv<float, V>(invstd + c0);
        mu2 = ldv<float, V>(mean2 + c0); is2 = ldv<float, V>(invstd2 + c0);
    }
    __device__ void operator()(long r, int c0, float (&o)[3][V]) const {
        fvec<V> g = ldv<YT, V>(dy + r * dy_pitch + c0);
        const fvec<V> xv = ldv<XT, V>(x + r * x_pitch + c0), xw = ldv<XT, V>(x2 + r * x2_pitch + c0);
        const unsigned char *mp = mask + r * (C >> 2) + (c0 >> 2);
#pragma unroll
        for (int h = 0; h < V / 4; ++h) {
            const unsigned m = mp[h];
#pragma unroll
            for (int e = 0; e < 4; ++e) g.q[h][e] = (m >> e) & 1u ? g.q[h][e] : 0.f;
        }
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const int h = e >> 2, k = e & 3;
            o[0][e] = g.q[h][k];
            o[1][e] = g.q[h][k] * ((xv.q[h][k] - mu.q[h][k]) * is.q[h][k]);
            o[2][e] = g.q[h][k] * ((xw.q[h][k] - mu2.q[h][k]) * is2.q[h][k]);
        }
    }
};

// colreduce2_kernel with three accumulators; partial rows are [4C] = [s0 | s1 | s0 | s2]
template <class F, int V>
__global__ void __launch_bounds__(RED_THREADS) colreduce3_kernel(F f, long rows, int C, int cvB, int RG, long rows_per_chunk,
                                                                 double *__restrict__ partial, Sweep sw = Sweep{0, 1, 0}) {
    __shared__ double red[RED_THREADS * 3 * V];
    const int tid = threadIdx.x;
    const int cv = tid % cvB, rg = tid / cvB;
    const int c0 = (blockIdx.y * cvB + cv) * V;
    const long r0 = (long)blockIdx.x * rows_per_chunk;
    const long r1 = min(rows, r0 + rows_per_chunk);
    double s[3][V];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int e = 0; e < V; ++e) s[k][e] = 0;
    if (rg < RG && c0 < C) {
        f.init(c0);
        auto take = [&](long r) {
            float o[3][V];
            f(r, c0, o);
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
                for (int e = 0; e < V; ++e) s[k][e] += (double)o[k][e];
        };
        if (sw.mode) {
            const long lend = sw.positions(rows), lstep = (long)gridDim.x * RG;
            for (long l = (long)blockIdx.x * RG + rg; l < lend; l += lstep) {
                const long r = sw.row(l);
                if (r < rows) take(r);
            }
        } else {
            for (long r = r0 + rg; r < r1; r += RG) take(r);
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int e = 0; e < V; ++e) red[(tid * 3 + k) * V + e] = s[k][e];
    __syncthreads();
    if (rg == 0 && c0 < C) {
        for (int g = 1; g < RG; ++g)
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
                for (int e = 0; e < V; ++e) s[k][e] += red[((g * cvB + cv) * 3 + k) * V + e];
        double *dst = partial + (long)blockIdx.x * 4 * C;
#pragma unroll
        for (int e = 0; e < V; ++e) {
            dst[c0 + e] = s[0][e];
            dst[C + c0 + e] = s[1][e];
            dst[2 * C + c0 + e] = s[0][e];
            dst[3 * C + c0 + e] = s[2][e];
        }
    }
}

__global__ void bn_finalize_kernel(const double *__restrict__ sums, double count, int C, float eps, float momentum,
                                   float *__restrict__ mean, float *__restrict__ invstd, float *__restrict__ rmean,
                                   float *__restrict__ rvar) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double m = sums[c] / count;
    double var = sums[C + c] / count - m * m;
    if (var < 0) var = 0;
    mean[c] = (float)m;
    invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (rmean) rmean[c] = (1.f - momentum) * rmean[c] + momentum * (float)m;
    if (rvar) {
        const double unbiased = count > 1 ? var * (count / (count - 1.0)) : var;
        rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unbiased;
    }
}

// partial [chunks][2C] (sum | sum of squares per conv row tile) -> sums[2C] and, with fin.on, the batch-norm constants of
// bn_finalize_kernel plus `num_batches_tracked += 1`, in the launch that finishes the reduction.  Until round 5 long lists of
// row tiles were first cut into G groups by partial_sum_groups_kernel (more workgroups than C/32 read) and this kernel added
// the G group rows: two launches of ~5 us each, 54 times per step, on the forward critical path.  Round 6: ONE launch for
// lists of 256 .. 1023 row tiles -- what the reduction costs is dependent load rounds per thread, so a workgroup takes fewer
// channels (CH = 8 instead of 32) and more row slices (32 instead of 8): at most ~8 rounds of four loads in flight per thread,
// C / 8 workgroups; longer lists keep the groups launch (rcf_sum_partials_bn).  (A single-launch form of ANOTHER kind -- the workgroup that draws the last ticket of its channels adds the group rows --
// was measured at 37 us per call against 20 for the two launches: the agent-scope release fence every workgroup needs
// before its ticket writes back an L2 full of the conv's dirty output.)
struct FinArgs {
    double count;
    float eps, momentum;
    float *mean, *invstd, *rmean, *rvar;
    long long *nbt;
    int on;
};

template <int CH>       // channels per workgroup: 512 threads = 2 halves x CH channels x (256 / CH) row slices
__global__ void __launch_bounds__(512) sum_finalize_kernel(const double *__restrict__ partial, int chunks, int C,
                                                           double *__restrict__ out, FinArgs fin) {
    // threads 0-255 add the sums, 256-511 the sums of squares of the block's CH channels (SL row slices each)
    constexpr int SL = 256 / CH;
    __shared__ double sh[2][256];
    const int h = threadIdx.x >> 8, tid = threadIdx.x & 255;
    const int j = tid % CH, sl = tid / CH;
    const int c = blockIdx.x * CH + j;
    const int k0 = 0, k1 = chunks;
    const long n = 2L * C;
    {
        double acc = 0;
        if (c < C) {
            const long col = (long)h * C + c;
            double a0 = 0, a1 = 0, a2 = 0, a3 = 0;        // four independent chains, combined in a fixed order
            int k = k0 + sl;
            for (; k + 3 * SL < k1; k += 4 * SL) {
                a0 += partial[(long)k * n + col];
                a1 += partial[(long)(k + SL) * n + col];
                a2 += partial[(long)(k + 2 * SL) * n + col];
                a3 += partial[(long)(k + 3 * SL) * n + col];
            }
            for (; k < k1; k += SL) a0 += partial[(long)k * n + col];
            acc = (a0 + a1) + (a2 + a3);
        }
        sh[h][tid] = acc;
    }
    __syncthreads();
    double t = 0;                                          // threads with sl == 0: the block's sum for (half h, channel c)
    if (sl == 0) {
#pragma unroll
        for (int q = 0; q < SL; ++q) t += sh[h][q * CH + j];
    }
    __syncthreads();
    if (sl == 0) sh[h][j] = t;                             // hand both halves to the thread that finalizes the channel
    __syncthreads();
    if (h != 0 || sl != 0 || c >= C) return;
    const double t0 = sh[0][j], t1 = sh[1][j];
    if (out) {
        out[c] = t0;
        out[C + c] = t1;
    }
    if (fin.on) {                                          // bn_finalize_kernel's arithmetic
        const double m = t0 / fin.count;
        double var = t1 / fin.count - m * m;
        if (var < 0) var = 0;
        fin.mean[c] = (float)m;
        fin.invstd[c] = (float)(1.0 / sqrt(var + (double)fin.eps));
        if (fin.rmean) fin.rmean[c] = (1.f - fin.momentum) * fin.rmean[c] + fin.momentum * (float)m;
        if (fin.rvar) {
            const double unbiased = fin.count > 1 ? var * (fin.count / (fin.count - 1.0)) : var;
            fin.rvar[c] = (1.f - fin.momentum) * fin.rvar[c] + fin.momentum * (float)unbiased;
        }
        if (c == 0 && fin.nbt) fin.nbt[0] += 1;
    }
}

__global__ void invstd_from_var_kernel(const float *__restrict__ var, int C, float eps, float *__restrict__ invstd) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) invstd[c] = 1.0f / sqrtf(var[c] + eps);
}

// Element-wise passes.  A thread keeps ONE vector of V channels for the whole kernel (its per-channel constants live in
// registers: no reloads, no index divisions in the loop) and walks the rows with a grid stride, two rows in flight.
// Thread layout: cvt = min(CV, 256) threads across the channels, 256 / cvt rows per block; layers wider than 256
// vectors loop over channel chunks (grid.y).
// out = mask ? dy : 0 -- the gradient a batch norm + ReLU join passes to its identity branch, materialised (the fallback of the
// deferred form: csrc/igemm_conv.hip adds the masked tensor in the data gradient's epilogue instead, rcf_conv2d_dgrad_add_f32)
template <typename T>
__global__ void __launch_bounds__(256) relu_mask_copy_kernel(const T *__restrict__ dy, int dy_pitch, const unsigned char *__restrict__ mask,
                                                             T *__restrict__ out, int out_pitch, long rows, int C, int beta) {
    const int CV = C >> 2;
    const long n = rows * CV;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const long r = i / CV;
        const int c0 = (int)(i - r * CV) << 2;
        fvec<4> g = ldv<T, 4>(dy + r * dy_pitch + c0);
        const unsigned m = mask[r * CV + (c0 >> 2)];
#pragma unroll
        for (int e = 0; e < 4; ++e) g.q[0][e] = (m >> e) & 1u ? g.q[0][e] : 0.f;
        if (beta) {
            const fvec<4> o = ldv<T, 4>(out + r * out_pitch + c0);
#pragma unroll
            for (int e = 0; e < 4; ++e) g.q[0][e] += o.q[0][e];
        }
        stv<T, 4>(out + r * out_pitch + c0, g);
    }
}

struct EwGeom {
    int cvt, rpb;      // threads across channels, rows per block
    dim3 grid;
};
EwGeom ew_geom(long rows, int CV) {
    EwGeom g;
    g.cvt = CV < 256 ? CV : 256;
    g.rpb = 256 / g.cvt;
    const int cchunks = (CV + g.cvt - 1) / g.cvt;
    long rb = (rows + g.rpb - 1) / g.rpb;              // row blocks if every block took one group of rows
    long bx = rb / 8 > 0 ? rb / 8 : 1;                  // >= 8 row groups per block (2 in flight at a time)
    const long cap = 4096 / cchunks > 0 ? 4096 / cchunks : 1;
    if (bx > cap) bx = cap;
    g.grid = dim3((unsigned)bx, (unsigned)cchunks);
    return g;
}

// ---- fp16 pair planes (include/rcf_hip.h RCF_CONV_X_PLANES): x * 2^k = h + m, [pixel][h: C fp16 | m: C fp16]
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int pl_exponent(unsigned bound_bits) {          // csrc/igemm_conv.hip h2_exponent: the consumers' rule
    const int e = (int)((bound_bits >> 23) & 0xffu);
    if (bound_bits == 0u) return 0;
    const int k = 14 - (e - 127);
    return k > 100 ? 100 : (k < -100 ? -100 : k);
}
__device__ __forceinline__ float pl_pow2(int k) { return __builtin_bit_cast(float, (unsigned)(127 + k) << 23); }
// h = fp16(v s) (s a power of two: the product is exact), m = fp16(v s - h) (the difference is exact in fp32): the values the
// conv kernels' own split (split2h) produces
__device__ __forceinline__ void pl_store4(char *pix, int C, int c0, const f32x4 v, float s) {
    f16x4 h, m;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float t = v[e] * s;
        h[e] = (_Float16)t;
        m[e] = (_Float16)(t - (float)h[e]);
    }
    *reinterpret_cast<f16x4 *>(pix + 2 * c0) = h;
    *reinterpret_cast<f16x4 *>(pix + 2 * C + 2 * c0) = m;
}
// block-wide max of a non-negative float over all threads (every thread calls it; result in every thread).  Safe to call
// back to back: the trailing barrier keeps a fast wave's next store out of the scratch a slow wave is still reading.
__device__ __forceinline__ float block_max_f(float v) {
    __shared__ float sh_bm[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    if ((threadIdx.x & 63) == 0) sh_bm[threadIdx.x >> 6] = v;
    __syncthreads();
    const float r = fmaxf(fmaxf(sh_bm[0], sh_bm[1]), fmaxf(sh_bm[2], sh_bm[3]));
    __syncthreads();
    return r;
}

// PL: 0 y only; 1 y and its pair planes; 2 the pair planes only (the output's only consumers are convs).  The planes' scale comes
// from an upper bound every workgroup derives from the per-channel constants before it reads a single element:
//     |y_c| <= |gamma_c| invstd_c (max|x| + |mean_c|) + |beta_c| (+ max|residual|)
// (max|x|: the conv epilogue's range of x; ReLU only lowers it).  Workgroup 0 leaves the bound in *amax: the consumers scale by it.
// rn: the residual is itself a conv output in front of a (train-mode, already finalized) batch norm without ReLU -- a stage's
// downsample branch, models/resnet.py:293-294 -- and is normalised on the fly: that norm's apply pass (a read and a write of
// the 4C-wide tensor) does not exist.  Same operations in the same order as the pass would have used.
struct ResNorm {
    const float *mean = nullptr, *invstd = nullptr, *gamma = nullptr, *beta = nullptr;
};

template <typename XT, typename YT, int V, int PL = 0>
__global__ void __launch_bounds__(256) bn_apply_kernel(const XT *__restrict__ x, int x_pitch,
                                                       const YT *__restrict__ res, int r_pitch,
                                                       YT *__restrict__ y, int y_pitch, long rows, int C, int cvt, int rpb,
                                                       const float *__restrict__ mean, const float *__restrict__ invstd,
                                                       const float *__restrict__ gamma, const float *__restrict__ beta,
                                                       int relu, const float *__restrict__ scale, long rows_per_image,
                                                       unsigned char *__restrict__ mask,
                                                       unsigned *__restrict__ amax, Sweep sw, char *__restrict__ planes = nullptr,
                                                       const unsigned *__restrict__ amax_x = nullptr,
                                                       const unsigned *__restrict__ amax_res = nullptr, ResNorm rn = ResNorm{}) {
    const int CV = C / V;
    const int cx = threadIdx.x % cvt, ry = threadIdx.x / cvt;
    const int cv = blockIdx.y * cvt + cx;
    const bool active = ry < rpb && cv < CV;
    const int c0 = cv * V;
    unsigned mx = 0u;
    float pls = 1.f;                                   // 2^k of the planes
    if constexpr (PL != 0) {
        const float ax = __uint_as_float(*amax_x);
        float ar = res ? __uint_as_float(*amax_res) : 0.f;
        float b = 0.f;
        for (int c = threadIdx.x; c < C; c += 256)
            b = fmaxf(b, fabsf(gamma[c]) * invstd[c] * (ax + fabsf(mean[c])) + fabsf(beta[c]));
        if (res && rn.mean) {                          // amax_res is the range of the residual BEFORE its norm
            float br = 0.f;
            for (int c = threadIdx.x; c < C; c += 256)
                br = fmaxf(br, fabsf(rn.gamma[c]) * rn.invstd[c] * (ar + fabsf(rn.mean[c])) + fabsf(rn.beta[c]));
            ar = block_max_f(br) * 1.0000005f;
        }
        b = (block_max_f(b) + ar) * 1.0000005f;        // the elements are rounded at every step of their own evaluation
        const unsigned bits = __float_as_uint(b);
        if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *amax = bits;
        pls = pl_pow2(pl_exponent(bits));
    }
    if (active) {
        const fvec<V> mu = ldv<float, V>(mean + c0), is = ldv<float, V>(invstd + c0);
        const fvec<V> ga = ldv<float, V>(gamma + c0), be = ldv<float, V>(beta + c0);
        const long rstep = (long)gridDim.x * rpb;
        const bool rnorm = res && rn.mean;
        fvec<V> rmu{}, ris{}, rga{}, rbe{};
        if (rnorm) {
            rmu = ldv<float, V>(rn.mean + c0); ris = ldv<float, V>(rn.invstd + c0);
            rga = ldv<float, V>(rn.gamma + c0); rbe = ldv<float, V>(rn.beta + c0);
        }
        auto finish = [&](long r, const fvec<V> &xv, const fvec<V> &rv_in) {
            fvec<V> o, rv = rv_in;
#pragma unroll
            for (int e = 0; e < V; ++e) {
                const int h = e >> 2, k = e & 3;
                // (x - mean) * invstd * gamma + beta, evaluated in the reference's order
                o.q[h][k] = (xv.q[h][k] - mu.q[h][k]) * is.q[h][k] * ga.q[h][k] + be.q[h][k];
            }
            if (rnorm) {
#pragma unroll
                for (int e = 0; e < V; ++e) {
                    const int h = e >> 2, k = e & 3;
                    rv.q[h][k] = (rv.q[h][k] - rmu.q[h][k]) * ris.q[h][k] * rga.q[h][k] + rbe.q[h][k];
                    if constexpr (!std::is_same<YT, float>::value) rv.q[h][k] = (float)(YT)rv.q[h][k];   // as the stored tensor was
                }
            }
            if (res) {
#pragma unroll
                for (int h = 0; h < V / 4; ++h) o.q[h] += rv.q[h];
            }
            if (relu) {
#pragma unroll
                for (int h = 0; h < V / 4; ++h) {
                    if (mask)
                        mask[r * (C >> 2) + (c0 >> 2) + h] = (unsigned char)((o.q[h][0] > 0.f) | ((o.q[h][1] > 0.f) << 1) |
                                                                             ((o.q[h][2] > 0.f) << 2) | ((o.q[h][3] > 0.f) << 3));
#pragma unroll
                    for (int e = 0; e < 4; ++e) o.q[h][e] = o.q[h][e] > 0.f ? o.q[h][e] : 0.f;
                }
            }
            if (scale) {
                const fvec<V> sc = ldv<float, V>(scale + (r / rows_per_image) * C + c0);
#pragma unroll
                for (int h = 0; h < V / 4; ++h) o.q[h] *= sc.q[h];
            }
            if constexpr (PL != 2) stv<YT, V>(y + r * y_pitch + c0, o);
            if constexpr (PL != 0) {
#pragma unroll
                for (int h = 0; h < V / 4; ++h) pl_store4(planes + r * (4L * C), C, c0 + 4 * h, o.q[h], pls);
            } else {
#pragma unroll
                for (int e = 0; e < V; ++e) mx = max(mx, __float_as_uint(fabsf(o.q[e >> 2][e & 3])));
            }
        };
        // l: position in the sweep (ascending in time), sw.row(l): the row it stands for (>= rows: a padding position)
        long l = (long)blockIdx.x * rpb + ry;
        const long lend = sw.positions(rows);
        for (; l + rstep < lend; l += 2 * rstep) {      // two rows in flight
            const long ra = sw.row(l), rb = sw.row(l + rstep);
            const bool va = ra < rows, vb = rb < rows;
            fvec<V> x0{}, x1{}, r0{}, r1{};
            if (va) x0 = ldv<XT, V>(x + ra * x_pitch + c0);
            if (vb) x1 = ldv<XT, V>(x + rb * x_pitch + c0);
            if (res) {
                if (va) r0 = ldv<YT, V>(res + ra * r_pitch + c0);
                if (vb) r1 = ldv<YT, V>(res + rb * r_pitch + c0);
            }
            if (va) finish(ra, x0, r0);
            if (vb) finish(rb, x1, r1);
        }
        if (l < lend) {
            const long ra = sw.row(l);
            if (ra < rows) {
                const fvec<V> x0 = ldv<XT, V>(x + ra * x_pitch + c0);
                fvec<V> r0{};
                if (res) r0 = ldv<YT, V>(res + ra * r_pitch + c0);
                finish(ra, x0, r0);
            }
        }
    }
    if constexpr (PL == 0) {
        if (amax) block_amax(mx, amax);
    }
}

// PL = 1: dx is written as fp16 pair planes (its only consumers are the conv's data and weight gradient).  Bound, per channel, from
// constants known before the pass:   |dx_c| <= |gamma_c| invstd_c (max|g| + |mean g|_c + X_c |mean g xhat|_c),
// X_c = (max|x| + |mean_c|) invstd_c >= max |xhat_c|, max|g| <= max|dy| max(chan_scale) (the ReLU mask only lowers it; the
// Dropout2d scale, 0 or 1 / keep per (image, channel), enters with its largest entry: `nscale` values are scanned in the
// prologue).  Workgroup 0 leaves the bound in *amax.
template <typename XT, typename YT, int V, int PL = 0>
__global__ void __launch_bounds__(256) bn_bwd_apply_kernel(
    const YT *__restrict__ dy, int dy_pitch, const XT *__restrict__ x, int x_pitch, const YT *__restrict__ y,
    int y_pitch, XT *__restrict__ dx, int dx_pitch, YT *__restrict__ dres, int dres_pitch, int res_beta,
    long rows, int C, int cvt, int rpb, const float *__restrict__ mean, const float *__restrict__ invstd,
    const float *__restrict__ gamma, int relu, const float *__restrict__ scale, long rows_per_image,
    const double *__restrict__ sums2, const double *__restrict__ sums2_param, double count,
    float *__restrict__ dgamma, float *__restrict__ dbeta, const unsigned char *__restrict__ mask,
    unsigned *__restrict__ amax, Sweep sw, const unsigned *__restrict__ amax_x = nullptr,
    const unsigned *__restrict__ amax_dy = nullptr, long nscale = 0) {
    const int CV = C / V;
    float pls = 1.f;
    if constexpr (PL != 0) {
        float smax = 1.f;
        if (scale) {
            float m = 0.f;
            for (long i = threadIdx.x; i < nscale; i += 256) m = fmaxf(m, fabsf(scale[i]));
            smax = block_max_f(m);
            __syncthreads();                           // block_max_f's scratch is reused below
        }
        const float ax = __uint_as_float(*amax_x), ag = __uint_as_float(*amax_dy) * smax;
        const float ic = (float)(1.0 / count);
        float b = 0.f;
        for (int c = threadIdx.x; c < C; c += 256) {
            const float is = invstd[c], X = (ax + fabsf(mean[c])) * is;
            b = fmaxf(b, fabsf(gamma[c]) * is * (ag + fabsf((float)sums2[c] * ic) + X * fabsf((float)sums2[C + c] * ic)));
        }
        b = block_max_f(b) * 1.0000005f;
        const unsigned bits = __float_as_uint(b);
        if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *amax = bits;
        pls = pl_pow2(pl_exponent(bits));
    }
    if (blockIdx.x == 0 && blockIdx.y == 0) {
        // parameter gradients come from THIS rank's sums: the data-parallel gradient all-reduce adds the ranks
        for (int c = threadIdx.x; c < C; c += blockDim.x) {
            if (dgamma) dgamma[c] += (float)sums2_param[C + c];
            if (dbeta) dbeta[c] += (float)sums2_param[c];
        }
    }
    const int cx = threadIdx.x % cvt, ry = threadIdx.x / cvt;
    const int cv = blockIdx.y * cvt + cx;
    const bool active = ry < rpb && cv < CV;
    const int c0 = cv * V;
    unsigned mx = 0u;
    if (active) {
        const float inv_count = (float)(1.0 / count);
        const fvec<V> mu = ldv<float, V>(mean + c0), is = ldv<float, V>(invstd + c0), ga = ldv<float, V>(gamma + c0);
        fvec<V> sg, sgx;
#pragma unroll
        for (int e = 0; e < V; ++e) {
            sg.q[e >> 2][e & 3] = (float)sums2[c0 + e] * inv_count;
            sgx.q[e >> 2][e & 3] = (float)sums2[C + c0 + e] * inv_count;
        }
        const long rstep = (long)gridDim.x * rpb;
        auto finish = [&](long r, fvec<V> g, const fvec<V> &xv) {
            if (scale) {
                const fvec<V> sc = ldv<float, V>(scale + (r / rows_per_image) * C + c0);
#pragma unroll
                for (int h = 0; h < V / 4; ++h) g.q[h] *= sc.q[h];
            }
            if (relu && mask) {
                const unsigned char *mp = mask + r * (C >> 2) + (c0 >> 2);
#pragma unroll
                for (int h = 0; h < V / 4; ++h) {
                    const unsigned m = mp[h];
#pragma unroll
                    for (int e = 0; e < 4; ++e) g.q[h][e] = (m >> e) & 1u ? g.q[h][e] : 0.f;
                }
            } else if (relu) {
                const fvec<V> yv = ldv<YT, V>(y + r * y_pitch + c0);
#pragma unroll
                for (int e = 0; e < V; ++e) g.q[e >> 2][e & 3] = yv.q[e >> 2][e & 3] > 0.f ? g.q[e >> 2][e & 3] : 0.f;
            }
            fvec<V> o;
#pragma unroll
            for (int e = 0; e < V; ++e) {
                const int h = e >> 2, k = e & 3;
                const float xh = (xv.q[h][k] - mu.q[h][k]) * is.q[h][k];
                o.q[h][k] = ga.q[h][k] * is.q[h][k] * (g.q[h][k] - sg.q[h][k] - xh * sgx.q[h][k]);
            }
            if constexpr (PL != 0) {
#pragma unroll
                for (int h = 0; h < V / 4; ++h) pl_store4(reinterpret_cast<char *>(dx) + r * (4L * C), C, c0 + 4 * h, o.q[h], pls);
            } else {
                stv<XT, V>(dx + r * dx_pitch + c0, o);
#pragma unroll
                for (int e = 0; e < V; ++e) mx = max(mx, __float_as_uint(fabsf(o.q[e >> 2][e & 3])));
            }
            if (dres) {
                YT *dr = dres + r * dres_pitch + c0;
                if (res_beta) {
                    const fvec<V> old = ldv<YT, V>(dr);
#pragma unroll
                    for (int h = 0; h < V / 4; ++h) g.q[h] += old.q[h];
                }
                stv<YT, V>(dr, g);
            }
        };
        long l = (long)blockIdx.x * rpb + ry;           // position in the sweep (see Sweep)
        const long lend = sw.positions(rows);
        for (; l + rstep < lend; l += 2 * rstep) {      // two rows in flight
            const long ra = sw.row(l), rb = sw.row(l + rstep);
            const bool va = ra < rows, vb = rb < rows;
            fvec<V> g0{}, g1{}, x0{}, x1{};
            if (va) g0 = ldv<YT, V>(dy + ra * dy_pitch + c0);
            if (vb) g1 = ldv<YT, V>(dy + rb * dy_pitch + c0);
            if (va) x0 = ldv<XT, V>(x + ra * x_pitch + c0);
            if (vb) x1 = ldv<XT, V>(x + rb * x_pitch + c0);
            if (va) finish(ra, g0, x0);
            if (vb) finish(rb, g1, x1);
        }
        if (l < lend) {
            const long ra = sw.row(l);
            if (ra < rows) finish(ra, ldv<YT, V>(dy + ra * dy_pitch + c0), ldv<XT, V>(x + ra * x_pitch + c0));
        }
    }
    if constexpr (PL == 0) {
        if (amax) block_amax(mx, amax);
    }
}

// ---- ... and ONE apply pass for both norms (rcf_bn_bwd_apply2_mp): dy and the sign bits are read once, dx = the gradient of
// conv3's output and dx2 = that of the downsample conv's come out together (fp32, or both as fp16 pair planes: PL = 1, each with
// its own bound).  Per element the operations of bn_bwd_apply_kernel in its order.
struct BwdSecond {
    const void *x;
    void *dx;
    int x_pitch, dx_pitch;
    const float *mean, *invstd, *gamma;
    const double *sums2, *sums2_param;
    float *dgamma, *dbeta;
    unsigned *amax;
    const unsigned *amax_x;
};

template <typename XT, typename YT, int V, int PL = 0>
__global__ void __launch_bounds__(256) bn_bwd_apply2_kernel(
    const YT *__restrict__ dy, int dy_pitch, const XT *__restrict__ x, int x_pitch, XT *__restrict__ dx, int dx_pitch, long rows,
    int C, int cvt, int rpb, const float *__restrict__ mean, const float *__restrict__ invstd, const float *__restrict__ gamma,
    const double *__restrict__ sums2, const double *__restrict__ sums2_param, double count, float *__restrict__ dgamma,
    float *__restrict__ dbeta, const unsigned char *__restrict__ mask, unsigned *__restrict__ amax, Sweep sw,
    const unsigned *__restrict__ amax_x, const unsigned *__restrict__ amax_dy, BwdSecond b2) {
    const int CV = C / V;
    const XT *__restrict__ x2 = reinterpret_cast<const XT *>(b2.x);
    XT *__restrict__ dx2 = reinterpret_cast<XT *>(b2.dx);
    float pls = 1.f, pls2 = 1.f;
    if constexpr (PL != 0) {
        const float ag = __uint_as_float(*amax_dy), ic = (float)(1.0 / count);
        const float ax = __uint_as_float(*amax_x), ax2 = __uint_as_float(*b2.amax_x);
        float b = 0.f, bb = 0.f;
        for (int c = threadIdx.x; c < C; c += 256) {
            const float is = invstd[c], X = (ax + fabsf(mean[c])) * is;
            b = fmaxf(b, fabsf(gamma[c]) * is * (ag + fabsf((float)sums2[c] * ic) + X * fabsf((float)sums2[C + c] * ic)));
            const float is2 = b2.invstd[c], X2 = (ax2 + fabsf(b2.mean[c])) * is2;
            bb = fmaxf(bb, fabsf(b2.gamma[c]) * is2 * (ag + fabsf((float)b2.sums2[c] * ic) + X2 * fabsf((float)b2.sums2[C + c] * ic)));
        }
        b = block_max_f(b) * 1.0000005f;
        __syncthreads();
        bb = block_max_f(bb) * 1.0000005f;
        const unsigned bits = __float_as_uint(b), bits2 = __float_as_uint(bb);
        if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) { *amax = bits; *b2.amax = bits2; }
        pls = pl_pow2(pl_exponent(bits));
        pls2 = pl_pow2(pl_exponent(bits2));
    }
    if (blockIdx.x == 0 && blockIdx.y == 0) {
        for (int c = threadIdx.x; c < C; c += blockDim.x) {
            if (dgamma) dgamma[c] += (float)sums2_param[C + c];
            if (dbeta) dbeta[c] += (float)sums2_param[c];
            if (b2.dgamma) b2.dgamma[c] += (float)b2.sums2_param[C + c];
            if (b2.dbeta) b2.dbeta[c] += (float)b2.sums2_param[c];
        }
    }
    const int cx = threadIdx.x % cvt, ry = threadIdx.x / cvt;
    const int cv = blockIdx.y * cvt + cx;
    const bool active = ry < rpb && cv < CV;
    const int c0 = cv * V;
    unsigned mx = 0u, mx2 = 0u;
    if (active) {
        const float inv_count = (float)(1.0 / count);
        const fvec<V> mu = ldv<float, V>(mean + c0), is = ldv<float, V>(invstd + c0), ga = ldv<float, V>(gamma + c0);
        const fvec<V> mu2 = ldv<float, V>(b2.mean + c0), is2 = ldv<float, V>(b2.invstd + c0), ga2 = ldv<float, V>(b2.gamma + c0);
        fvec<V> sg, sgx, sg2, sgx2;
#pragma unroll
        for (int e = 0; e < V; ++e) {
            sg.q[e >> 2][e & 3] = (float)sums2[c0 + e] * inv_count;
            sgx.q[e >> 2][e & 3] = (float)sums2[C + c0 + e] * inv_count;
            sg2.q[e >> 2][e & 3] = (float)b2.sums2[c0 + e] * inv_count;
            sgx2.q[e >> 2][e & 3] = (float)b2.sums2[C + c0 + e] * inv_count;
        }
        const long rstep = (long)gridDim.x * rpb;
        auto finish = [&](long r, fvec<V> g, const fvec<V> &xv, const fvec<V> &xw) {
            const unsigned char *mp = mask + r * (C >> 2) + (c0 >> 2);
#pragma unroll
            for (int h = 0; h < V / 4; ++h) {
                const unsigned m = mp[h];
#pragma unroll
                for (int e = 0; e < 4; ++e) g.q[h][e] = (m >> e) & 1u ? g.q[h][e] : 0.f;
            }
            fvec<V> o, o2;
#pragma unroll
            for (int e = 0; e < V; ++e) {
                const int h = e >> 2, k = e & 3;
                const float xh = (xv.q[h][k] - mu.q[h][k]) * is.q[h][k];
                o.q[h][k] = ga.q[h][k] * is.q[h][k] * (g.q[h][k] - sg.q[h][k] - xh * sgx.q[h][k]);
                const float xh2 = (xw.q[h][k] - mu2.q[h][k]) * is2.q[h][k];
                o2.q[h][k] = ga2.q[h][k] * is2.q[h][k] * (g.q[h][k] - sg2.q[h][k] - xh2 * sgx2.q[h][k]);
            }
            if constexpr (PL != 0) {
#pragma unroll
                for (int h = 0; h < V / 4; ++h) {
                    pl_store4(reinterpret_cast<char *>(dx) + r * (4L * C), C, c0 + 4 * h, o.q[h], pls);
                    pl_store4(reinterpret_cast<char *>(dx2) + r * (4L * C), C, c0 + 4 * h, o2.q[h], pls2);
                }
            } else {
                stv<XT, V>(dx + r * dx_pitch + c0, o);
                stv<XT, V>(dx2 + r * b2.dx_pitch + c0, o2);
#pragma unroll
                for (int e = 0; e < V; ++e) {
                    mx = max(mx, __float_as_uint(fabsf(o.q[e >> 2][e & 3])));
                    mx2 = max(mx2, __float_as_uint(fabsf(o2.q[e >> 2][e & 3])));
                }
            }
        };
        long l = (long)blockIdx.x * rpb + ry;
        const long lend = sw.positions(rows);
        for (; l + rstep < lend; l += 2 * rstep) {      // two rows in flight
            const long ra = sw.row(l), rb = sw.row(l + rstep);
            const bool va = ra < rows, vb = rb < rows;
            fvec<V> g0{}, g1{}, x0{}, x1{}, w0{}, w1{};
            if (va) { g0 = ldv<YT, V>(dy + ra * dy_pitch + c0); x0 = ldv<XT, V>(x + ra * x_pitch + c0); w0 = ldv<XT, V>(x2 + ra * b2.x_pitch + c0); }
            if (vb) { g1 = ldv<YT, V>(dy + rb * dy_pitch + c0); x1 = ldv<XT, V>(x + rb * x_pitch + c0); w1 = ldv<XT, V>(x2 + rb * b2.x_pitch + c0); }
            if (va) finish(ra, g0, x0, w0);
            if (vb) finish(rb, g1, x1, w1);
        }
        if (l < lend) {
            const long ra = sw.row(l);
            if (ra < rows) finish(ra, ldv<YT, V>(dy + ra * dy_pitch + c0), ldv<XT, V>(x + ra * x_pitch + c0), ldv<XT, V>(x2 + ra * b2.x_pitch + c0));
        }
    }
    if constexpr (PL == 0) {
        if (amax) block_amax(mx, amax);
        if (b2.amax) { __syncthreads(); block_amax(mx2, b2.amax); }
    }
}

template <typename XT, int V>
struct ColsumOp {
    const XT *x;
    int pitch;
    __device__ void init(int) {}
    __device__ void operator()(long r, int c0, float (&a)[V], float (&b)[V]) const {
        const fvec<V> v = ldv<XT, V>(x + r * pitch + c0);
#pragma unroll
        for (int e = 0; e < V; ++e) {
            a[e] = v.q[e >> 2][e & 3];
            b[e] = 0.f;
        }
    }
};

__global__ void __launch_bounds__(256) colsum_final_kernel(const double *__restrict__ partial, int chunks, int C,
                                                           float *__restrict__ out, int beta) {
    __shared__ double sh[256];
    const int j = threadIdx.x & 31, s = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + j;
    double acc = 0;
    if (c < C)
        for (int k = s; k < chunks; k += 8) acc += partial[(long)k * 2 * C + c];
    sh[threadIdx.x] = acc;
    __syncthreads();
    if (s == 0 && c < C) {
        double t = 0;
#pragma unroll
        for (int q = 0; q < 8; ++q) t += sh[q * 32 + j];
        out[c] = (beta ? out[c] : 0.f) + (float)t;
    }
}

inline int ew_blocks(long total) {
    long b = (total + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

}  // namespace

// partial [chunks][2C] -> sums[2C] (may be NULL with fin) and, with fin, the batch-norm constants; one or two launches
int rcf_sum_partials_bn(const double *partial, int chunks, int C, double *sums, double *scratch,
                        const rcf_bn_finalize *fin, void *stream) {
    if (!partial || chunks <= 0 || C <= 0 || (!sums && !fin)) return RCF_EINVAL;
    if (fin && (!fin->mean || !fin->invstd || !(fin->count > 0))) return RCF_EINVAL;
    hipStream_t st = rcf_stream(stream);
    FinArgs fa{};
    if (fin) {
        fa.count = fin->count; fa.eps = fin->eps; fa.momentum = fin->momentum;
        fa.mean = fin->mean; fa.invstd = fin->invstd; fa.rmean = fin->running_mean; fa.rvar = fin->running_var;
        fa.nbt = fin->num_batches_tracked; fa.on = 1;
    }
    if (scratch && chunks >= 1024) {
        // very long lists (the 3210 row tiles of layer1 at 8 pairs of 480x854): 64 groups of rows first -- the one-launch form
        // with 4 channels per workgroup measured 21 us there against 10 + a launch boundary for the two launches
        const int G = 64, per = rcf_cdiv(chunks, G);
        hipLaunchKernelGGL(partial_sum_groups_kernel, dim3(rcf_cdiv(2 * C, 32), G), dim3(256), 0, st, partial, chunks, per,
                           2 * C, scratch);
        RCF_LAUNCH_CHECK();
        partial = scratch;
        chunks = G;
    }
    // many row tiles (256 .. 1023): fewer channels and more row slices per workgroup, so that a thread walks at most ~32 rows:
    // 7.3 us in ONE launch against 5.3 + 4.8 us in two (round 6 step profile, 41 of the 54 forward reductions)
    if (chunks >= 256) hipLaunchKernelGGL(sum_finalize_kernel<8>, dim3(rcf_cdiv(C, 8)), dim3(512), 0, st, partial, chunks, C, sums, fa);
    else hipLaunchKernelGGL(sum_finalize_kernel<32>, dim3(rcf_cdiv(C, 32)), dim3(512), 0, st, partial, chunks, C, sums, fa);
    RCF_LAUNCH_CHECK();
    return 0;
}

extern "C" size_t rcf_bn_stats_workspace_bytes(long rows, int C) {
    if (rows <= 0 || C <= 0 || C % 4) return 0;
    const ColGeom g = col_geom(rows, C);
    int chunks = g.chunks;
    if (C % 8 == 0) {                                   // the 8-channel (bf16) geometry may cut the rows into more chunks
        const ColGeom g8 = col_geom(rows, C, 8);
        chunks = g8.chunks > chunks ? g8.chunks : chunks;
    }
    return (size_t)chunks * 2 * C * sizeof(double);
}

namespace {
// bf16 on both sides and C % 8 == 0: 8 channels (one 16-byte access) per thread
inline int vec_width(int xdt, int ydt, int C, int pa, int pb, int pc, int pd) {
    return (xdt == RCF_BF16 && ydt == RCF_BF16 && C % 8 == 0 && pa % 8 == 0 && pb % 8 == 0 && pc % 8 == 0 && pd % 8 == 0) ? 8 : 4;
}
}  // namespace

extern "C" int rcf_bn_stats_mp(const void *x, int xdt, long rows, int C, int pitch, double *sums, void *workspace,
                               size_t workspace_bytes, void *stream) {
    if (!x || !sums || rows <= 0 || C <= 0 || C % 4 || pitch % 4 || pitch < C) return RCF_EINVAL;
    if (!workspace || workspace_bytes < rcf_bn_stats_workspace_bytes(rows, C)) return RCF_EWORKSPACE;
    const int V = vec_width(xdt, xdt, C, pitch, 8, 8, 8);
    const ColGeom g = col_geom(rows, C, V);
    hipStream_t st = rcf_stream(stream);
    if (V == 8) {
        StatsOp<bf16_t, 8> op{(const bf16_t *)x, pitch};
        hipLaunchKernelGGL((colreduce2_kernel<StatsOp<bf16_t, 8>, 8>), dim3(g.chunks, g.cgroups), dim3(RED_THREADS), 0, st,
                           op, rows, C, g.cvB, g.RG, g.rows_per_chunk, (double *)workspace);
    } else {
#define RCF_CALL(XT)                                                                                                     \
    StatsOp<XT, 4> op{(const XT *)x, pitch};                                                                             \
    hipLaunchKernelGGL((colreduce2_kernel<StatsOp<XT, 4>, 4>), dim3(g.chunks, g.cgroups), dim3(RED_THREADS), 0, st, op, \
                       rows, C, g.cvB, g.RG, g.rows_per_chunk, (double *)workspace)
        RCF_DISPATCH1(xdt, RCF_CALL);
#undef RCF_CALL
    }
    RCF_LAUNCH_CHECK();
    launch_partial_sum((const double *)workspace, g.chunks, 2 * C, sums, st);
    RCF_LAUNCH_CHECK();
    return 0;
}

extern "C" int rcf_bn_stats_f32(const float *x, long rows, int C, int pitch, double *sums, void *workspace,
                                size_t workspace_bytes, void *stream) {
    return rcf_bn_stats_mp(x, RCF_F32, rows, C, pitch, sums, workspace, workspace_bytes, stream);
}

extern "C" int rcf_sum_partials_f64(const double *partial, int chunks, int n, double *out, double *scratch,
                                    void *stream) {
    if (!partial || !out || chunks <= 0 || n <= 0) return RCF_EINVAL;
    hipStream_t st = rcf_stream(stream);
    if (scratch && chunks >= 256) {
        // many partial rows (one per conv row tile): G groups of rows first, so that more than n/32 workgroups read
        const int G = chunks >= 2048 ? 64 : 16, per = rcf_cdiv(chunks, G);
        hipLaunchKernelGGL(partial_sum_groups_kernel, dim3(rcf_cdiv(n, 32), G), dim3(256), 0, st, partial, chunks, per, n,
                           scratch);
        RCF_LAUNCH_CHECK();
        partial = scratch;
        chunks = G;
    }
    launch_partial_sum(partial, chunks, n, out, st);
    RCF_LAUNCH_CHECK();
    return 0;
}

extern "C" int rcf_bn_finalize_f32(const double *sums, double count, int C, float eps, float momentum, float *mean,
                                   float *invstd, float *running_mean, float *running_var, void *stream) {
    if (!sums || !mean || !invstd || C <= 0 || count <= 0) return RCF_EINVAL;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(rcf_cdiv(C, 256)), dim3(256), 0, rcf_stream(stream), sums, count, C,
                       eps, momentum, mean, invstd, running_mean, running_var);
    RCF_LAUNCH_CHECK();
    return 0;
}

extern "C" int rcf_bn_invstd_from_var_f32(const float *var, int C, float eps, float *invstd, void *stream) {
    if (!var || !invstd || C <= 0) return RCF_EINVAL;
    hipLaunchKernelGGL(invstd_from_var_kernel, dim3(rcf_cdiv(C, 256)), dim3(256), 0, rcf_stream(stream), var, C, eps,
                       invstd);
    RCF_LAUNCH_CHECK();
    return 0;
}

extern "C" int rcf_bn_apply_mp(const void *x, int xdt, int x_pitch, const void *residual, int r_pitch, void *y, int ydt,
                               int y_pitch, long rows, int C, const float *mean, const float *invstd,
                               const float *gamma, const float *beta, int relu, const float *chan_scale,
                               long rows_per_image, unsigned char *relu_mask, unsigned *amax_out, void *planes_out,
                               const unsigned *amax_x, const unsigned *amax_res, unsigned flags, void *stream) {
    return rcf_bn_apply_res_mp(x, xdt, x_pitch, residual, r_pitch, nullptr, y, ydt, y_pitch, rows, C, mean, invstd, gamma, beta, relu,
                               chan_scale, rows_per_image, relu_mask, amax_out, planes_out, amax_x, amax_res, flags, stream);
}

extern "C" int rcf_bn_apply_res_mp(const void *x, int xdt, int x_pitch, const void *residual, int r_pitch,
                                   const rcf_bn_res_norm *res_norm, void *y, int ydt, int y_pitch, long rows, int C,
                                   const float *mean, const float *invstd, const float *gamma, const float *beta, int relu,
                                   const float *chan_scale, long rows_per_image, unsigned char *relu_mask,
                                   unsigned *amax_out, void *planes_out, const unsigned *amax_x, const unsigned *amax_res,
                                   unsigned flags, void *stream) {
    ResNorm rn{};
    if (res_norm) {
        if (!residual || !res_norm->mean || !res_norm->invstd || !res_norm->gamma || !res_norm->beta) return RCF_EINVAL;
        rn.mean = res_norm->mean; rn.invstd = res_norm->invstd; rn.gamma = res_norm->gamma; rn.beta = res_norm->beta;
    }
    const bool planes_only = planes_out && (flags & RCF_BN_Y_PLANES_ONLY);
    if (!x || (!y && !planes_only) || !mean || !invstd || !gamma || !beta || rows <= 0 || C <= 0 || C % 4) return RCF_EINVAL;
    if (x_pitch % 4 || y_pitch % 4 || (residual && r_pitch % 4)) return RCF_EINVAL;
    if (chan_scale && rows_per_image <= 0) return RCF_EINVAL;
    if (planes_out) {
        // pair planes: fp32 tensors, contiguous pixels of C % 8 == 0 channels, the input's range (and the residual's), a slot for
        // the bound; no per-channel dropout scale (its maximum would have to enter the bound)
        if (xdt != RCF_F32 || ydt != RCF_F32 || C % 8 || !amax_x || !amax_out || (residual && !amax_res) || chan_scale ||
            !rcf_aligned16(planes_out))
            return RCF_EINVAL;
        const EwGeom g = ew_geom(rows, C / 4);
        const Sweep sw = make_sweep(2, rows, g.rpb, (long)C * 4, flags);
#define RCF_CALL(PLv)                                                                                                     \
    hipLaunchKernelGGL((bn_apply_kernel<float, float, 4, PLv>), g.grid, dim3(256), 0, rcf_stream(stream), (const float *)x, x_pitch, \
                       (const float *)residual, r_pitch, (float *)y, y_pitch, rows, C, g.cvt, g.rpb, mean, invstd, gamma, beta, relu, \
                       (const float *)nullptr, 1L, relu_mask, amax_out, sw, (char *)planes_out, amax_x, amax_res, rn)
        if (planes_only) RCF_CALL(2);
        else RCF_CALL(1);
#undef RCF_CALL
        RCF_LAUNCH_CHECK();
        return 0;
    }
    if (vec_width(xdt, ydt, C, x_pitch, y_pitch, residual ? r_pitch : 8, 8) == 8) {
        const EwGeom g = ew_geom(rows, C / 8);
        hipLaunchKernelGGL((bn_apply_kernel<bf16_t, bf16_t, 8>), g.grid, dim3(256), 0,
                           rcf_stream(stream), (const bf16_t *)x, x_pitch, (const bf16_t *)residual, r_pitch, (bf16_t *)y,
                           y_pitch, rows, C, g.cvt, g.rpb, mean, invstd, gamma, beta, relu, chan_scale,
                           rows_per_image > 0 ? rows_per_image : 1, relu_mask, amax_out, make_sweep(2, rows, g.rpb, (long)C * 2, flags),
                           (char *)nullptr, (const unsigned *)nullptr, (const unsigned *)nullptr, rn);
    } else {
        const EwGeom g = ew_geom(rows, C / 4);
#define RCF_CALL(XT, YT)                                                                                                 \
    hipLaunchKernelGGL((bn_apply_kernel<XT, YT, 4>), g.grid, dim3(256), 0, rcf_stream(stream),                          \
                       (const XT *)x, x_pitch, (const YT *)residual, r_pitch, (YT *)y, y_pitch, rows, C, g.cvt, g.rpb,   \
                       mean, invstd, gamma, beta, relu, chan_scale, rows_per_image > 0 ? rows_per_image : 1, relu_mask,  \
                       amax_out, make_sweep(2, rows, g.rpb, (long)C * (xdt == RCF_BF16 ? 2 : 4), flags),               \
                       (char *)nullptr, (const unsigned *)nullptr, (const unsigned *)nullptr, rn)
        RCF_DISPATCH2(xdt, ydt, RCF_CALL);
#undef RCF_CALL
    }
    RCF_LAUNCH_CHECK();
    return 0;
}

extern "C" int rcf_bn_apply_f32(const float *x, int x_pitch, const float *residual, int r_pitch, float *y,
                                int y_pitch, long rows, int C, const float *mean, const float *invstd,
                                const float *gamma, const float *beta, int relu, const float *chan_scale,
                                long rows_per_image, unsigned char *relu_mask, unsigned *amax_out, void *stream) {
    return rcf_bn_apply_mp(x, RCF_F32, x_pitch, residual, r_pitch, y, RCF_F32, y_pitch, rows, C, mean, invstd, gamma, beta,
                           relu, chan_scale, rows_per_image, relu_mask, amax_out, nullptr, nullptr, nullptr, 0u, stream);
}

extern "C" int rcf_bn_bwd_reduce_mp(const void *dy, int ydt, int dy_pitch, const void *x, int xdt, int x_pitch,
                                    const void *y, int y_pitch, long rows, int C, const float *mean,
                                    const float *invstd, int relu, const unsigned char *relu_mask,
                                    const float *chan_scale, long rows_per_image, double *sums2, void *workspace,
                                    size_t workspace_bytes, unsigned flags, void *stream) {
    if (!dy || !x || !mean || !invstd || !sums2 || rows <= 0 || C <= 0 || C % 4) return RCF_EINVAL;
    if (relu && !y && !relu_mask) return RCF_EINVAL;
    if (dy_pitch % 4 || x_pitch % 4 || (relu && !relu_mask && y_pitch % 4)) return RCF_EINVAL;
    if (!workspace || workspace_bytes < rcf_bn_stats_workspace_bytes(rows, C)) return RCF_EWORKSPACE;
    const int V = vec_width(xdt, ydt, C, dy_pitch, x_pitch, (relu && !relu_mask) ? y_pitch : 8, 8);
    const ColGeom g = col_geom(rows, C, V);
    hipStream_t st = rcf_stream(stream);
    if (V == 8) {
        BwdOp<bf16_t, bf16_t, 8> op{(const bf16_t *)dy, (const bf16_t *)x, (const bf16_t *)y, mean, invstd, chan_scale,
                                    dy_pitch, x_pitch, y_pitch, relu, C, rows_per_image > 0 ? rows_per_image : 1, relu_mask, {}, {}};
        hipLaunchKernelGGL((colreduce2_kernel<BwdOp<bf16_t, bf16_t, 8>, 8>), dim3(g.chunks, g.cgroups), dim3(RED_THREADS), 0,
                           st, op, rows, C, g.cvB, g.RG, g.rows_per_chunk, (double *)workspace, make_sweep(2, rows, g.RG, (long)C * 2, flags));
    } else {
#define RCF_CALL(XT, YT)                                                                                                    \
    BwdOp<XT, YT, 4> op{(const YT *)dy, (const XT *)x, (const YT *)y, mean, invstd, chan_scale, dy_pitch, x_pitch, y_pitch, \
                        relu, C, rows_per_image > 0 ? rows_per_image : 1, relu_mask, {}, {}};                               \
    hipLaunchKernelGGL((colreduce2_kernel<BwdOp<XT, YT, 4>, 4>), dim3(g.chunks, g.cgroups), dim3(RED_THREADS), 0, st, op,  \
                       rows, C, g.cvB, g.RG, g.rows_per_chunk, (double *)workspace, make_sweep(2, rows, g.RG, (long)C * (xdt == RCF_BF16 ? 2 : 4), flags))
        RCF_DISPATCH2(xdt, ydt, RCF_CALL);
#undef RCF_CALL
    }
    RCF_LAUNCH_CHECK();
    launch_partial_sum((const double *)workspace, g.chunks, 2 * C, sums2, st);
    RCF_LAUNCH_CHECK();
    return 0;
}

extern "C" int rcf_bn_bwd_reduce_f32(const float *dy, int dy_pitch, const float *x, int x_pitch, const float *y,
                                     int y_pitch, long rows, int C, const float *mean, const float *invstd, int relu,
                                     const unsigned char *relu_mask, const float *chan_scale, long rows_per_image,
                                     double *sums2, void *workspace, size_t workspace_bytes, void *stream) {
    return rcf_bn_bwd_reduce_mp(dy, RCF_F32, dy_pitch, x, RCF_F32, x_pitch, y, y_pitch, rows, C, mean, invstd, relu,
                                relu_mask, chan_scale, rows_per_image, sums2, workspace, workspace_bytes, 0u, stream);
}

extern "C" int rcf_bn_bwd_apply_mp(const void *dy, int ydt, int dy_pitch, const void *x, int xdt, int x_pitch,
                                   const void *y, int y_pitch, void *dx, int dx_pitch, void *dres, int dres_pitch,
                                   int res_beta, long rows, int C, const float *mean, const float *invstd,
                                   const float *gamma, int relu, const unsigned char *relu_mask,
                                   const float *chan_scale, long rows_per_image, const double *sums2,
                                   const double *sums2_local, double count, float *dgamma, float *dbeta,
                                   unsigned *amax_out, const unsigned *amax_x, const unsigned *amax_dy, unsigned flags,
                                   void *stream) {
    if (!dy || !x || !dx || !mean || !invstd || !gamma || !sums2 || rows <= 0 || C <= 0 || C % 4 || count <= 0)
        return RCF_EINVAL;
    if (relu && !y && !relu_mask) return RCF_EINVAL;
    if (dy_pitch % 4 || x_pitch % 4 || dx_pitch % 4 || (dres && dres_pitch % 4)) return RCF_EINVAL;
    if (flags & RCF_BN_DX_PLANES) {
        // dx as fp16 pair planes (see rcf_bn_apply_mp): fp32 tensors, contiguous dx pixels, the ranges of x and dy, a slot for the bound
        if (xdt != RCF_F32 || ydt != RCF_F32 || C % 8 || dx_pitch != C || !amax_x || !amax_dy || !amax_out ||
            !rcf_aligned16(dx) || (relu && !relu_mask) || (chan_scale && rows % (rows_per_image > 0 ? rows_per_image : 1)))
            return RCF_EINVAL;
        const EwGeom g = ew_geom(rows, C / 4);
        hipLaunchKernelGGL((bn_bwd_apply_kernel<float, float, 4, 1>), g.grid, dim3(256), 0, rcf_stream(stream), (const float *)dy,
                           dy_pitch, (const float *)x, x_pitch, (const float *)y, y_pitch, (float *)dx, dx_pitch, (float *)dres,
                           dres_pitch, res_beta, rows, C, g.cvt, g.rpb, mean, invstd, gamma, relu, chan_scale,
                           rows_per_image > 0 ? rows_per_image : 1, sums2, sums2_local ? sums2_local : sums2, count, dgamma, dbeta,
                           relu_mask, amax_out, make_sweep(1, rows, g.rpb, (long)C * 4, flags), amax_x, amax_dy,
                           chan_scale ? (rows / (rows_per_image > 0 ? rows_per_image : 1)) * (long)C : 0L);
        RCF_LAUNCH_CHECK();
        return 0;
    }
    if (vec_width(xdt, ydt, C, dy_pitch, x_pitch, dx_pitch, dres ? dres_pitch : 8) == 8 &&
        (!(relu && !relu_mask) || y_pitch % 8 == 0)) {
        const EwGeom g = ew_geom(rows, C / 8);
        hipLaunchKernelGGL((bn_bwd_apply_kernel<bf16_t, bf16_t, 8>), g.grid, dim3(256), 0,
                           rcf_stream(stream), (const bf16_t *)dy, dy_pitch, (const bf16_t *)x, x_pitch, (const bf16_t *)y,
                           y_pitch, (bf16_t *)dx, dx_pitch, (bf16_t *)dres, dres_pitch, res_beta, rows, C, g.cvt, g.rpb, mean, invstd,
                           gamma, relu, chan_scale, rows_per_image > 0 ? rows_per_image : 1, sums2,
                           sums2_local ? sums2_local : sums2, count, dgamma, dbeta, relu_mask, amax_out, make_sweep(1, rows, g.rpb, (long)C * 2, flags));
    } else {
        const EwGeom g = ew_geom(rows, C / 4);
#define RCF_CALL(XT, YT)                                                                                                     \
    hipLaunchKernelGGL((bn_bwd_apply_kernel<XT, YT, 4>), g.grid, dim3(256), 0, rcf_stream(stream),                          \
                       (const YT *)dy, dy_pitch, (const XT *)x, x_pitch, (const YT *)y, y_pitch, (XT *)dx, dx_pitch,         \
                       (YT *)dres, dres_pitch, res_beta, rows, C, g.cvt, g.rpb, mean, invstd, gamma, relu, chan_scale,       \
                       rows_per_image > 0 ? rows_per_image : 1, sums2, sums2_local ? sums2_local : sums2, count, dgamma,     \
                       dbeta, relu_mask, amax_out, make_sweep(1, rows, g.rpb, (long)C * (xdt == RCF_BF16 ? 2 : 4), flags))
        RCF_DISPATCH2(xdt, ydt, RCF_CALL);
#undef RCF_CALL
    }
    RCF_LAUNCH_CHECK();
    return 0;
}

extern "C" int rcf_bn_bwd_apply_f32(const float *dy, int dy_pitch, const float *x, int x_pitch, const float *y,
                                    int y_pitch, float *dx, int dx_pitch, float *dres, int dres_pitch, int res_beta,
                                    long rows, int C, const float *mean, const float *invstd, const float *gamma,
                                    int relu, const unsigned char *relu_mask, const float *chan_scale,
                                    long rows_per_image, const double *sums2, const double *sums2_local, double count,
                                    float *dgamma, float *dbeta, unsigned *amax_out, void *stream) {
    return rcf_bn_bwd_apply_mp(dy, RCF_F32, dy_pitch, x, RCF_F32, x_pitch, y, y_pitch, dx, dx_pitch, dres, dres_pitch,
                               res_beta, rows, C, mean, invstd, gamma, relu, relu_mask, chan_scale, rows_per_image, sums2,
                               sums2_local, count, dgamma, dbeta, amax_out, nullptr, nullptr, 0u, stream);
}

extern "C" int rcf_relu_mask_copy_mp(const void *dy, int dt, int dy_pitch, const unsigned char *relu_mask, void *out, int out_pitch,
                                     long rows, int C, int beta, void *stream) {
    if (!dy || !relu_mask || !out || rows <= 0 || C <= 0 || C % 4 || dy_pitch % 4 || out_pitch % 4 || dy_pitch < C || out_pitch < C)
        return RCF_EINVAL;
    const long n = rows * (C / 4);
    const int blocks = (int)((n + 255) / 256 > 8192 ? 8192 : (n + 255) / 256);
#define RCF_CALL(T)                                                                                                  \
    hipLaunchKernelGGL((relu_mask_copy_kernel<T>), dim3(blocks), dim3(256), 0, rcf_stream(stream), (const T *)dy, dy_pitch, \
                       relu_mask, (T *)out, out_pitch, rows, C, beta)
    RCF_DISPATCH1(dt, RCF_CALL);
#undef RCF_CALL
    RCF_LAUNCH_CHECK();
    return 0;
}

extern "C" int rcf_colsum_mp(const void *x, int xdt, long rows, int C, int pitch, float *out, int beta, void *workspace,
                             size_t workspace_bytes, void *stream) {
    if (!x || !out || rows <= 0 || C <= 0 || C % 4 || pitch % 4 || pitch < C) return RCF_EINVAL;
    if (!workspace || workspace_bytes < rcf_bn_stats_workspace_bytes(rows, C)) return RCF_EWORKSPACE;
    const ColGeom g = col_geom(rows, C);
    hipStream_t st = rcf_stream(stream);
#define RCF_CALL(XT)                                                                                                      \
    ColsumOp<XT, 4> op{(const XT *)x, pitch};                                                                             \
    hipLaunchKernelGGL((colreduce2_kernel<ColsumOp<XT, 4>, 4>), dim3(g.chunks, g.cgroups), dim3(RED_THREADS), 0, st, op, \
                       rows, C, g.cvB, g.RG, g.rows_per_chunk, (double *)workspace)
    RCF_DISPATCH1(xdt, RCF_CALL);
#undef RCF_CALL
    RCF_LAUNCH_CHECK();
    hipLaunchKernelGGL(colsum_final_kernel, dim3(rcf_cdiv(C, 32)), dim3(256), 0, st, (const double *)workspace,
                       g.chunks, C, out, beta);
    RCF_LAUNCH_CHECK();
    return 0;
}

extern "C" int rcf_colsum_f32(const float *x, long rows, int C, int pitch, float *out, int beta, void *workspace,
                              size_t workspace_bytes, void *stream) {
    return rcf_colsum_mp(x, RCF_F32, rows, C, pitch, out, beta, workspace, workspace_bytes, stream);
}


/* ---- a stage's first block: the join and its (lazy) downsample norm in one reduction and one apply pass (include/rcf_hip.h) */
extern "C" int rcf_bn_bwd_reduce2_mp(const void *dy, int ydt, int dy_pitch, const void *x, int xdt, int x_pitch, const void *x2,
                                     int x2_pitch, long rows, int C, const float *mean, const float *invstd, const float *mean2,
                                     const float *invstd2, const unsigned char *relu_mask, double *sums4, void *workspace,
                                     size_t workspace_bytes, unsigned flags, void *stream) {
    if (!dy || !x || !x2 || !mean || !invstd || !mean2 || !invstd2 || !relu_mask || !sums4 || rows <= 0 || C <= 0 || C % 4) return RCF_EINVAL;
    if (dy_pitch % 4 || x_pitch % 4 || x2_pitch % 4 || xdt != ydt || (xdt != RCF_F32 && xdt != RCF_BF16)) return RCF_EINVAL;
    if (!workspace || workspace_bytes < 2 * rcf_bn_stats_workspace_bytes(rows, C)) return RCF_EWORKSPACE;
    const int V = (xdt == RCF_BF16 && C % 8 == 0 && dy_pitch % 8 == 0 && x_pitch % 8 == 0 && x2_pitch % 8 == 0) ? 8 : 4;
    const ColGeom g = col_geom(rows, C, V);
    hipStream_t st = rcf_stream(stream);
    if (xdt == RCF_BF16 && V == 8) {
        BwdOp2<bf16_t, bf16_t, 8> op{(const bf16_t *)dy, (const bf16_t *)x, (const bf16_t *)x2, mean, invstd, mean2, invstd2, dy_pitch,
                                     x_pitch, x2_pitch, C, relu_mask, {}, {}, {}, {}};
        hipLaunchKernelGGL((colreduce3_kernel<BwdOp2<bf16_t, bf16_t, 8>, 8>), dim3(g.chunks, g.cgroups), dim3(RED_THREADS), 0, st, op,
                           rows, C, g.cvB, g.RG, g.rows_per_chunk, (double *)workspace, make_sweep(2, rows, g.RG, (long)C * 2, flags));
    } else if (xdt == RCF_BF16) {
        BwdOp2<bf16_t, bf16_t, 4> op{(const bf16_t *)dy, (const bf16_t *)x, (const bf16_t *)x2, mean, invstd, mean2, invstd2, dy_pitch,
                                     x_pitch, x2_pitch, C, relu_mask, {}, {}, {}, {}};
        hipLaunchKernelGGL((colreduce3_kernel<BwdOp2<bf16_t, bf16_t, 4>, 4>), dim3(g.chunks, g.cgroups), dim3(RED_THREADS), 0, st, op,
                           rows, C, g.cvB, g.RG, g.rows_per_chunk, (double *)workspace, make_sweep(2, rows, g.RG, (long)C * 2, flags));
    } else {
        BwdOp2<float, float, 4> op{(const float *)dy, (const float *)x, (const float *)x2, mean, invstd, mean2, invstd2, dy_pitch,
                                   x_pitch, x2_pitch, C, relu_mask, {}, {}, {}, {}};
        hipLaunchKernelGGL((colreduce3_kernel<BwdOp2<float, float, 4>, 4>), dim3(g.chunks, g.cgroups), dim3(RED_THREADS), 0, st, op,
                           rows, C, g.cvB, g.RG, g.rows_per_chunk, (double *)workspace, make_sweep(2, rows, g.RG, (long)C * 4, flags));
    }
    RCF_LAUNCH_CHECK();
    launch_partial_sum((const double *)workspace, g.chunks, 4 * C, sums4, st);
    RCF_LAUNCH_CHECK();
    return 0;
}

extern "C" int rcf_bn_bwd_apply2_mp(const void *dy, int ydt, int dy_pitch, const void *x, int xdt, int x_pitch, void *dx,
                                    int dx_pitch, long rows, int C, const float *mean, const float *invstd, const float *gamma,
                                    const unsigned char *relu_mask, const double *sums2, const double *sums2_local, double count,
                                    float *dgamma, float *dbeta, unsigned *amax_out, const unsigned *amax_x,
                                    const unsigned *amax_dy, const rcf_bn_bwd_second *second, unsigned flags, void *stream) {
    if (!dy || !x || !dx || !mean || !invstd || !gamma || !sums2 || !relu_mask || !second || rows <= 0 || C <= 0 || C % 4 || count <= 0)
        return RCF_EINVAL;
    if (!second->x || !second->dx || !second->mean || !second->invstd || !second->gamma || !second->sums2) return RCF_EINVAL;
    if (dy_pitch % 4 || x_pitch % 4 || dx_pitch % 4 || second->x_pitch % 4 || second->dx_pitch % 4 || xdt != ydt) return RCF_EINVAL;
    BwdSecond b2{second->x, second->dx, second->x_pitch, second->dx_pitch, second->mean, second->invstd, second->gamma, second->sums2,
                 second->sums2_local ? second->sums2_local : second->sums2, second->dgamma, second->dbeta, second->amax_out, second->amax_x};
    const double *sp = sums2_local ? sums2_local : sums2;
    if (flags & RCF_BN_DX_PLANES) {
        if (xdt != RCF_F32 || C % 8 || dx_pitch != C || second->dx_pitch != C || !amax_x || !amax_dy || !amax_out || !second->amax_x ||
            !second->amax_out || !rcf_aligned16(dx) || !rcf_aligned16(second->dx))
            return RCF_EINVAL;
        const EwGeom g = ew_geom(rows, C / 4);
        hipLaunchKernelGGL((bn_bwd_apply2_kernel<float, float, 4, 1>), g.grid, dim3(256), 0, rcf_stream(stream), (const float *)dy, dy_pitch,
                           (const float *)x, x_pitch, (float *)dx, dx_pitch, rows, C, g.cvt, g.rpb, mean, invstd, gamma, sums2, sp, count,
                           dgamma, dbeta, relu_mask, amax_out, make_sweep(1, rows, g.rpb, (long)C * 4, flags), amax_x, amax_dy, b2);
    } else if (xdt == RCF_BF16 && C % 8 == 0 && dy_pitch % 8 == 0 && x_pitch % 8 == 0 && dx_pitch % 8 == 0 && second->x_pitch % 8 == 0 &&
               second->dx_pitch % 8 == 0) {
        const EwGeom g = ew_geom(rows, C / 8);
        hipLaunchKernelGGL((bn_bwd_apply2_kernel<bf16_t, bf16_t, 8, 0>), g.grid, dim3(256), 0, rcf_stream(stream), (const bf16_t *)dy,
                           dy_pitch, (const bf16_t *)x, x_pitch, (bf16_t *)dx, dx_pitch, rows, C, g.cvt, g.rpb, mean, invstd, gamma, sums2,
                           sp, count, dgamma, dbeta, relu_mask, amax_out, make_sweep(1, rows, g.rpb, (long)C * 2, flags), amax_x, amax_dy, b2);
    } else if (xdt == RCF_F32) {
        const EwGeom g = ew_geom(rows, C / 4);
        hipLaunchKernelGGL((bn_bwd_apply2_kernel<float, float, 4, 0>), g.grid, dim3(256), 0, rcf_stream(stream), (const float *)dy, dy_pitch,
                           (const float *)x, x_pitch, (float *)dx, dx_pitch, rows, C, g.cvt, g.rpb, mean, invstd, gamma, sums2, sp, count,
                           dgamma, dbeta, relu_mask, amax_out, make_sweep(1, rows, g.rpb, (long)C * 4, flags), amax_x, amax_dy, b2);
    } else {
        return RCF_EINVAL;
    }
    RCF_LAUNCH_CHECK();
    return 0;
}
