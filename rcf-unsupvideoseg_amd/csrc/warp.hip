// Backward bilinear warp by optical flow, forward-splat occlusion maps and the photometric residual.
// Reference: utils/warp_utils.py:84-94 (flow_warp = mesh_grid + norm_grid + grid_sample, align_corners
// True), :27-81,107-113 (get_occu_mask_backward), :97-104 (get_occu_mask_bidirection),
// models/amd/flow_loss.py:15-29 + models/amd/loss_blocks.py:46-65 (L1 + SSIM photometric loss).
//
// The reference makes ~6 passes over memory (two grids, normalise, permute, sample); here one thread
// per pixel reads the flow once and gathers the 4 taps per channel: 4*(2 + 2*C) algorithmic bytes per
// pixel (32 B for RGB), HBM-bound, coalesced along x.  Planar NCHW fp32 like the reference API.
// The normalise / un-normalise round trip of the reference is kept in the same float operations so
// sample positions round identically.
#include "rcf_common.h"

namespace {

inline int px_blocks(long total) {
    long b = (total + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 16384 ? 16384 : b));
}

struct Taps {
    int x0, y0;
    float wx, wy;     // fractional parts
    bool inx0, inx1, iny0, iny1;
    bool gx_live, gy_live;   // gradient wrt coordinate survives clipping
};

// pad_mode 0 = border, 1 = zeros
__device__ __forceinline__ Taps make_taps(float px, float py, int W, int H, int pad_mode) {
    // norm_grid (utils/warp_utils.py:17-24) then grid_sampler_unnormalize(align_corners=True)
    float gx = 2.0f * px / (float)(W - 1) - 1.0f;
    float gy = 2.0f * py / (float)(H - 1) - 1.0f;
    float ix = ((gx + 1.f) / 2.f) * (float)(W - 1);
    float iy = ((gy + 1.f) / 2.f) * (float)(H - 1);
    Taps t;
    t.gx_live = t.gy_live = true;
    if (pad_mode == 0) {
        if (!(ix > 0.f)) { ix = 0.f; t.gx_live = false; }
        else if (ix >= (float)(W - 1)) { ix = (float)(W - 1); t.gx_live = false; }
        if (!(iy > 0.f)) { iy = 0.f; t.gy_live = false; }
        else if (iy >= (float)(H - 1)) { iy = (float)(H - 1); t.gy_live = false; }
    }
    const float fx = floorf(ix), fy = floorf(iy);
    t.wx = ix - fx;
    t.wy = iy - fy;
    // huge |flow| -> keep the int conversion defined
    t.x0 = (int)fminf(fmaxf(fx, -2.f), (float)W + 1.f);
    t.y0 = (int)fminf(fmaxf(fy, -2.f), (float)H + 1.f);
    t.inx0 = t.x0 >= 0 && t.x0 < W;
    t.inx1 = t.x0 + 1 >= 0 && t.x0 + 1 < W;
    t.iny0 = t.y0 >= 0 && t.y0 < H;
    t.iny1 = t.y0 + 1 >= 0 && t.y0 + 1 < H;
    return t;
}

typedef float f32x2_a4 __attribute__((ext_vector_type(2), aligned(4)));   // 8-byte load, dword aligned

// Branch-free bilinear sample: the two horizontal taps of a row are adjacent, so each row is ONE 8-byte load at a
// clamped (always valid) position; taps that fall outside the image get weight 0 through selects, never through
// control flow -- all tap loads of a pixel (2 per channel) issue back to back.
struct Sampler {
    int off0, off1;          // float offsets of the two row loads inside a plane
    float w00, w01, w10, w11;   // weights of (row0: x0, x0+1), (row1: x0, x0+1); 0 for taps outside
    bool s00, s01;           // which element of the loaded pair is tap x0 / x0+1 (false = [0], true = [1])
};
__device__ __forceinline__ Sampler make_sampler(const Taps &t, int W, int H) {
    Sampler s;
    const int xb = min(max(t.x0, 0), W - 2);                 // pair base: both elements inside the row
    const int y0c = min(max(t.y0, 0), H - 1), y1c = min(max(t.y0 + 1, 0), H - 1);
    s.off0 = y0c * W + xb;
    s.off1 = y1c * W + xb;
    s.s00 = t.x0 != xb;                                       // x0 == xb + 1 (only when x0 == W-1)
    s.s01 = t.x0 + 1 != xb;                                   // false only when x0 == -1
    const float ex = 1.f - t.wx, ey = 1.f - t.wy;
    const float m0 = t.inx0 ? 1.f : 0.f, m1 = t.inx1 ? 1.f : 0.f, r0 = t.iny0 ? 1.f : 0.f, r1 = t.iny1 ? 1.f : 0.f;
    s.w00 = ey * ex * m0 * r0;
    s.w01 = ey * t.wx * m1 * r0;
    s.w10 = t.wy * ex * m0 * r1;
    s.w11 = t.wy * t.wx * m1 * r1;
    return s;
}
__device__ __forceinline__ float sample(const float *__restrict__ plane, const Sampler &s) {
    const f32x2_a4 a = *reinterpret_cast<const f32x2_a4 *>(plane + s.off0);
    const f32x2_a4 b = *reinterpret_cast<const f32x2_a4 *>(plane + s.off1);
    const float a0 = s.s00 ? a[1] : a[0], a1 = s.s01 ? a[1] : a[0];
    const float b0 = s.s00 ? b[1] : b[0], b1 = s.s01 ? b[1] : b[0];
    // same order of operations as grid_sample's accumulation: row 0 (x0, x0+1), then row 1
    float v = 0.f;
    v += a0 * s.w00 + a1 * s.w01;
    v += b0 * s.w10 + b1 * s.w11;
    return v;
}

// Workgroup -> pixels, XCD aware.  Workgroup id L runs on XCD L % 8, and each XCD has its own L2: an image is cut into 8
// horizontal bands and band k is walked, 512 pixels at a time, by the workgroups of XCD k only.  The source rows a
// run samples are sampled again by the runs above and below it; with runs dealt round-robin to the XCDs
// every L2 fetched its own copy of those rows (measured: 3.2x the algorithmic fetch bytes), now the same L2 serves them.
struct BandMap {
    int b, p0, p1;      // image, first pixel of this workgroup's run, end of its band
};
__device__ __forceinline__ BandMap band_map(int L, int H, int W, int runs_per_band) {
    const int band = L & 7, slot = L >> 3;
    const int rb = (H + 7) >> 3;                              // rows per band
    BandMap m;
    m.b = slot / runs_per_band;
    const int run = slot - m.b * runs_per_band;
    const int r0 = min(band * rb, H), r1 = min(r0 + rb, H);
    m.p0 = r0 * W + run * 512;
    m.p1 = r1 * W;
    return m;
}

__global__ void __launch_bounds__(256) flow_warp_kernel(const float *__restrict__ x, const float *__restrict__ flow,
                                                        float *__restrict__ out, int B, int C, int H, int W,
                                                        int pad_mode, int runs_per_band) {
    const int HW = H * W;
    const BandMap bm = band_map(blockIdx.x, H, W, runs_per_band);
    const long b = bm.b;
    const float *fl = flow + b * 2 * HW;
    const float *xb = x + b * C * HW;
    float *ob = out + b * C * HW;
    // a run is 512 pixels: two independent pixels per thread (twice the loads in flight per wavefront)
    const int pa = bm.p0 + threadIdx.x, pb = pa + 256;
    if (C == 3 && pb < bm.p1) {            // RGB: all twelve row loads in flight before the first blend
        const int ya = pa / W, xa = pa - ya * W, yb = pb / W, xb_ = pb - yb * W;
        const Sampler sa = make_sampler(make_taps((float)xa + fl[pa], (float)ya + fl[HW + pa], W, H, pad_mode), W, H);
        const Sampler sb = make_sampler(make_taps((float)xb_ + fl[pb], (float)yb + fl[HW + pb], W, H, pad_mode), W, H);
        float va[3], vb[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            va[c] = sample(xb + c * HW, sa);
            vb[c] = sample(xb + c * HW, sb);
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            ob[c * HW + pa] = va[c];
            ob[c * HW + pb] = vb[c];
        }
        return;
    }
    for (int p = pa; p < bm.p1 && p <= pb; p += 256) {
        const int py = p / W, px = p - py * W;
        const Sampler sm = make_sampler(make_taps((float)px + fl[p], (float)py + fl[HW + p], W, H, pad_mode), W, H);
        for (int c = 0; c < C; ++c) ob[c * HW + p] = sample(xb + c * HW, sm);
    }
}

__global__ void __launch_bounds__(256) flow_warp_bwd_kernel(const float *__restrict__ x,
                                                            const float *__restrict__ flow,
                                                            const float *__restrict__ dout, float *__restrict__ dx,
                                                            float *__restrict__ dflow, int B, int C, int H, int W,
                                                            int pad_mode) {
    const long HW = (long)H * W, total = (long)B * HW;
    const long step = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += step) {
        const long b = i / HW, p = i - b * HW;
        const int py = (int)(p / W), px = (int)(p - (long)py * W);
        const float fx = flow[(b * 2) * HW + p], fy = flow[(b * 2 + 1) * HW + p];
        const Taps t = make_taps((float)px + fx, (float)py + fy, W, H, pad_mode);
        const float ex = 1.f - t.wx, ey = 1.f - t.wy;
        float gix = 0.f, giy = 0.f;
        for (int c = 0; c < C; ++c) {
            const float g = dout[(b * C + c) * HW + p];
            const float *pl = x + (b * C + c) * HW;
            float *dpl = dx ? dx + (b * C + c) * HW : nullptr;
            float v00 = 0.f, v01 = 0.f, v10 = 0.f, v11 = 0.f;
            if (t.iny0 && t.inx0) { v00 = pl[(long)t.y0 * W + t.x0]; if (dpl) atomicAdd(dpl + (long)t.y0 * W + t.x0, g * ey * ex); }
            if (t.iny0 && t.inx1) { v01 = pl[(long)t.y0 * W + t.x0 + 1]; if (dpl) atomicAdd(dpl + (long)t.y0 * W + t.x0 + 1, g * ey * t.wx); }
            if (t.iny1 && t.inx0) { v10 = pl[(long)(t.y0 + 1) * W + t.x0]; if (dpl) atomicAdd(dpl + (long)(t.y0 + 1) * W + t.x0, g * t.wy * ex); }
            if (t.iny1 && t.inx1) { v11 = pl[(long)(t.y0 + 1) * W + t.x0 + 1]; if (dpl) atomicAdd(dpl + (long)(t.y0 + 1) * W + t.x0 + 1, g * t.wy * t.wx); }
            gix += g * (ey * (v01 - v00) + t.wy * (v11 - v10));
            giy += g * (ex * (v10 - v00) + t.wx * (v11 - v01));
        }
        if (dflow) {
            dflow[(b * 2) * HW + p] = t.gx_live ? gix : 0.f;
            dflow[(b * 2 + 1) * HW + p] = t.gy_live ? giy : 0.f;
        }
    }
}

// forward splat of the 4 bilinear weights (utils/warp_utils.py:27-81): corners that had to be clamped
// into the image are dropped.
__global__ void __launch_bounds__(256) splat_count_kernel(const float *__restrict__ flow, float *__restrict__ cnt,
                                                          int B, int H, int W) {
    const long HW = (long)H * W, total = (long)B * HW;
    const long step = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += step) {
        const long b = i / HW, p = i - b * HW;
        const int py = (int)(p / W), px = (int)(p - (long)py * W);
        const float x = (float)px + flow[(b * 2) * HW + p], y = (float)py + flow[(b * 2 + 1) * HW + p];
        const float x1 = floorf(x), y1 = floorf(y);
        const float xf = fminf(fmaxf(x1, 0.f), (float)(W - 1)), yf = fminf(fmaxf(y1, 0.f), (float)(H - 1));
        const float x0 = x1 + 1.f, y0 = y1 + 1.f;
        const float xc = fminf(fmaxf(x0, 0.f), (float)(W - 1)), yc = fminf(fmaxf(y0, 0.f), (float)(H - 1));
        const bool xco = x0 != xc, yco = y0 != yc, xfo = x1 != xf, yfo = y1 != yf;
        float *dst = cnt + b * HW;
        if (!(xco || yco)) atomicAdd(dst + (long)yc * W + (long)xc, (1.f - fabsf(x - xc)) * (1.f - fabsf(y - yc)));
        if (!(xco || yfo)) atomicAdd(dst + (long)yf * W + (long)xc, (1.f - fabsf(x - xc)) * (1.f - fabsf(y - yf)));
        if (!(xfo || yco)) atomicAdd(dst + (long)yc * W + (long)xf, (1.f - fabsf(x - xf)) * (1.f - fabsf(y - yc)));
        if (!(xfo || yfo)) atomicAdd(dst + (long)yf * W + (long)xf, (1.f - fabsf(x - xf)) * (1.f - fabsf(y - yf)));
    }
}

__global__ void __launch_bounds__(256) occ_threshold_kernel(const float *__restrict__ cnt, float *__restrict__ occ,
                                                            long n, float th) {
    const long step = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step)
        occ[i] = fminf(fmaxf(cnt[i], 0.f), 1.f) < th ? 1.f : 0.f;
}

__global__ void __launch_bounds__(256) occ_bidir_kernel(const float *__restrict__ f12, const float *__restrict__ f21,
                                                        float *__restrict__ occ, float scale, float bias, int B,
                                                        int H, int W) {
    const long HW = (long)H * W, total = (long)B * HW;
    const long step = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += step) {
        const long b = i / HW, p = i - b * HW;
        const int py = (int)(p / W), px = (int)(p - (long)py * W);
        const float ax = f12[(b * 2) * HW + p], ay = f12[(b * 2 + 1) * HW + p];
        const Sampler sm = make_sampler(make_taps((float)px + ax, (float)py + ay, W, H, 1), W, H);
        const float wx = sample(f21 + (b * 2) * HW, sm), wy = sample(f21 + (b * 2 + 1) * HW, sm);
        const float dx = ax + wx, dy = ay + wy;
        const float mag = (ax * ax + ay * ay) + (wx * wx + wy * wy);
        occ[i] = (dx * dx + dy * dy) > scale * mag + bias ? 1.f : 0.f;
    }
}

__device__ __forceinline__ void block_add2(double a, double b, double *out) {
    __shared__ double sh[8];
    a = wave_sum_d(a);
    b = wave_sum_d(b);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) { sh[w] = a; sh[4 + w] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(out, sh[0] + sh[1] + sh[2] + sh[3]);
        atomicAdd(out + 1, sh[4] + sh[5] + sh[6] + sh[7]);
    }
    __syncthreads();
}

__global__ void __launch_bounds__(256) warp_l1_kernel(const float *__restrict__ im1, const float *__restrict__ im2,
                                                      const float *__restrict__ flow, const float *__restrict__ occ,
                                                      double *__restrict__ out, int B, int C, int H, int W,
                                                      int pad_mode, int runs_per_band) {
    // XCD-aware bands as in flow_warp_kernel.  The (image, 512-pixel run) pairs of band k are dealt round-robin to the
    // gridDim.x / 8 workgroups of XCD k, so the runs in flight on an XCD at any time are neighbours (their source rows
    // share that L2) while the workgroup count -- each ends in two fp64 atomics -- stays small.
    const int HW = H * W;
    const int band = blockIdx.x & 7, j = blockIdx.x >> 3, Q = gridDim.x >> 3;
    const int rb = (H + 7) >> 3;
    const int r0 = min(band * rb, H), r1 = min(r0 + rb, H);
    const int pend = r1 * W;
    double s = 0, so = 0;
    long b = 0;
    const float *fl = flow;
    auto pixel = [&](int p, float &acc, float &o) {
        const int py = p / W, px = p - py * W;
        const Sampler sm = make_sampler(make_taps((float)px + fl[p], (float)py + fl[HW + p], W, H, pad_mode), W, H);
        o = occ ? occ[b * HW + p] : 1.f;
        acc = 0.f;
        if (C == 3) {                      // RGB: the nine loads in flight together
            float t[3], w[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                t[c] = im1[(b * 3 + c) * HW + p];
                w[c] = sample(im2 + (b * 3 + c) * HW, sm);
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) acc += fabsf(t[c] - w[c]);
        } else {
            for (int c = 0; c < C; ++c) acc += fabsf(im1[(b * C + c) * HW + p] - sample(im2 + (b * C + c) * HW, sm));
        }
    };
    const int R = B * runs_per_band;
    for (int r = j; r < R; r += Q) {
        b = r / runs_per_band;
        const int run = r - (int)b * runs_per_band;
        fl = flow + b * 2 * HW;
        const int p = r0 * W + run * 512 + threadIdx.x;
        // two independent pixels per trip: twice the loads in flight per wavefront
        if (p + 256 < pend) {
            float a0, o0, a1, o1;
            pixel(p, a0, o0);
            pixel(p + 256, a1, o1);
            s += (double)(a0 * o0);
            so += (double)o0;
            s += (double)(a1 * o1);
            so += (double)o1;
        } else if (p < pend) {
            float a0, o0;
            pixel(p, a0, o0);
            s += (double)(a0 * o0);
            so += (double)o0;
        }
    }
    block_add2(s, so, out);
}

// ---- RGB / border fast path ---------------------------------------------------------------------------------------------
// PMC on the per-pixel kernels above (profiles/r02_pmc_warp.txt): HBM delivers 1.03x the algorithmic bytes, yet the waves
// spend 41 % of their cycles in issue stalls with the vector ALU 51 % busy (182 VALU instructions per pixel: an integer
// division for (y, x), two IEEE divisions in the normalise round trip, selects for taps outside the image) and the
// texture addresser is the busiest unit (a 64-lane load costs ~20 of its cycles whether the lanes ask for 4 or for
// 4-byte-aligned 8 bytes).  The fast path keeps every float operation of the sample position and of the blend --
// results are bit-identical (tools/bench_warp.py, tests/test_kernels_gpu.py) -- and
//   * maps lanes to x inside 64x16 tiles (no per-pixel division), four independent pixels (rows) per thread;
//   * loads every tap as its own dword (lane addresses nearly consecutive: the 2-D tiles cut L2 requests by 20 %);
//   * computes x / (W-1) as  q = x*r;  q += fma(-q, d, x) * r  with r = RN(1/d) from the host -- correctly rounded
//     (Markstein: q is within 1 ulp and the residual is exact), 3 operations instead of the IEEE sequence's 10;
//   * uses that border mode clamps the position into [0, W-1] x [0, H-1]: the only tap outside is x0+1 = W (y0+1 = H) at
//     weight exactly 0, and folding it to (x0 = W-2, wx = 1) gives the same value with no masks and no selects;
//   * sends x and y through the position arithmetic as one float2 (v_pk_mul / v_pk_add / v_pk_fma).
// Measured on 64 frames of 480x854 (fused warp + L1): 262 -> 210 us, 3.6 -> 4.5 TB/s of algorithmic bytes.  Tried and
// slower: four consecutive pixels per thread with 16-byte stream loads (249 us; the strided pair gathers cost more cache
// accesses), staging the tile's source window in LDS (56 KB, two workgroups per CU, two barriers per tile: 421 us),
// non-temporal loads / stores on the streamed operands (+2 %), 2 or 8 rows per thread (223 / 211 us).  What bounds it
// now: the texture addresser, 88 % busy (18 dword loads per pixel = 72 requested bytes per 36 algorithmic ones).
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

struct WarpGeom {
    float dx, dy, rx, ry;       // W-1, H-1 and their correctly rounded reciprocals
};

struct Tap4 {
    int off;                    // offset of the (y0, x0) pair inside a plane; the (y0+1) pair is W further
    v2f w0, w1;                 // weights of (row y0: x0, x0+1), (row y0+1: x0, x0+1)
};

__device__ __forceinline__ Tap4 border_tap(int x, int y, float fx, float fy, const WarpGeom &g, int W, int H) {
    const v2f d = {g.dx, g.dy}, r = {g.rx, g.ry};
    const v2f p = {(float)x + fx, (float)y + fy};
    // norm_grid: 2*p/(size-1) - 1;  grid_sampler_unnormalize(align_corners=True): ((g+1)/2)*(size-1)
    const v2f t = 2.0f * p;
    v2f q = t * r;
    q = __builtin_elementwise_fma(__builtin_elementwise_fma(-q, d, t), r, q);        // == t / d, correctly rounded
    const v2f gn = q - 1.0f;
    v2f i = ((gn + 1.f) * 0.5f) * d;
    // clip_coordinates: below 0 (or NaN) -> 0, above size-1 -> size-1
    i.x = fminf(fmaxf(i.x, 0.f), g.dx);
    i.y = fminf(fmaxf(i.y, 0.f), g.dy);
    v2f f = {floorf(i.x), floorf(i.y)};
    v2f w = i - f;
    int x0 = (int)f.x, y0 = (int)f.y;
    if (x0 > W - 2) { x0 = W - 2; w.x = 1.f; }                 // tap x0+1 = W carried weight 0: same value from (W-2, 1)
    if (y0 > H - 2) { y0 = H - 2; w.y = 1.f; }
    const v2f e = 1.f - w;
    Tap4 tp;
    tp.off = y0 * W + x0;
    const v2f ew = {e.x, w.x};
    tp.w0 = e.y * ew;
    tp.w1 = w.y * ew;
    return tp;
}

__device__ __forceinline__ float blend(v2f a, v2f b, const Tap4 &tp) {
    const v2f ta = a * tp.w0, tb = b * tp.w1;
    return (ta.x + ta.y) + (tb.x + tb.y);
}

// XCD k (workgroup ids k mod 8) walks the k-th eighth of the row-major tile list of every image, so the source rows that
// neighbouring tiles share stay in one L2.  lane = x, wavefront wv takes rows wv, wv+4, ... of the tile.
constexpr int TILE_W = 64;
template <bool L1, int PX>
__global__ void __launch_bounds__(256) warp_rows_kernel(const float *__restrict__ im1, const float *__restrict__ im2,
                                                        const float *__restrict__ flow, const float *__restrict__ occ,
                                                        void *__restrict__ out_, int B, int H, int W, WarpGeom gm,
                                                        int tiles_x, int tiles_y) {
    const int HW = H * W;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3, Q = gridDim.x >> 3;
    const int ntiles = tiles_x * tiles_y;
    const int t0 = (int)((long)ntiles * xcd / 8), t1 = (int)((long)ntiles * (xcd + 1) / 8);
    const int per = t1 - t0, R = B * per;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    double s = 0, so = 0;
    for (int r = j; r < R; r += Q) {
        const int b = r / per, t = t0 + (r - b * per);
        const int ty = t / tiles_x, tx = t - ty * tiles_x;
        const float *fl = flow + (long)b * 2 * HW, *src = im2 + (long)b * 3 * HW;
        const int x = tx * TILE_W + lane;
        const int xc = min(x, W - 1);
        Tap4 tp[PX];
        int g[PX];
        bool ok[PX];
#pragma unroll
        for (int k = 0; k < PX; ++k) {
            const int y = ty * (4 * PX) + wv + 4 * k;
            ok[k] = x < W && y < H;
            const int yc = min(y, H - 1);
            g[k] = yc * W + xc;
            tp[k] = border_tap(xc, yc, fl[g[k]], fl[HW + g[k]], gm, W, H);
        }
        float v[3][PX];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float *pl = src + c * HW;
#pragma unroll
            for (int k = 0; k < PX; ++k) {
                const float *q = pl + tp[k].off;
                v[c][k] = blend(v2f{q[0], q[1]}, v2f{q[W], q[W + 1]}, tp[k]);
            }
        }
        if (L1) {
            const float *tgt = im1 + (long)b * 3 * HW;
            const float *oc = occ ? occ + (long)b * HW : nullptr;
#pragma unroll
            for (int k = 0; k < PX; ++k) {
                float acc = 0.f;
#pragma unroll
                for (int c = 0; c < 3; ++c) acc += fabsf(tgt[c * HW + g[k]] - v[c][k]);
                const float o = oc ? oc[g[k]] : 1.f;
                if (ok[k]) {
                    s += (double)(acc * o);
                    so += (double)o;
                }
            }
        } else {
            float *ob = (float *)out_ + (long)b * 3 * HW;
#pragma unroll
            for (int k = 0; k < PX; ++k)
                if (ok[k])
#pragma unroll
                    for (int c = 0; c < 3; ++c) ob[c * HW + g[k]] = v[c][k];
        }
    }
    if (L1) block_add2(s, so, (double *)out_);
}

// Even widths, fused warp + L1: a lane owns TWO horizontally adjacent pixels (128 x 8 tiles, 2 rows per thread): the streamed
// operands (flow, target, mask) arrive as aligned 8-byte loads -- less addresser work than two dword loads -- the taps stay
// dword gathers.  217 -> 210 us on 64 frames of 480x854, bit-identical; the plain warp (which also streams its output) is
// slower this way on wide images (265 vs 233 us) and keeps one pixel per lane.
template <bool L1>
__global__ void __launch_bounds__(256) warp_rows2_kernel(const float *__restrict__ im1, const float *__restrict__ im2,
                                                         const float *__restrict__ flow, const float *__restrict__ occ,
                                                         void *__restrict__ out_, int B, int H, int W, WarpGeom gm,
                                                         int tiles_x, int tiles_y) {
    const int HW = H * W;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3, Q = gridDim.x >> 3;
    const int ntiles = tiles_x * tiles_y;
    const int t0 = (int)((long)ntiles * xcd / 8), t1 = (int)((long)ntiles * (xcd + 1) / 8);
    const int per = t1 - t0, R = B * per;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    double s = 0, so = 0;
    for (int r = j; r < R; r += Q) {
        const int b = r / per, t = t0 + (r - b * per);
        const int ty = t / tiles_x, tx = t - ty * tiles_x;
        const float *fl = flow + (long)b * 2 * HW, *src = im2 + (long)b * 3 * HW;
        const int x = tx * 128 + 2 * lane;                          // pixels x, x + 1
        Tap4 tp[2][2];
        int g[2];
        bool ok[2][2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int y = ty * 8 + wv + 4 * k;
            const int yc = min(y, H - 1), xc = min(x, W - 2);       // W is even here: (x, x+1) both inside or both outside
            g[k] = yc * W + xc;
            ok[k][0] = ok[k][1] = x + 1 < W && y < H;
            const v2f fx = *reinterpret_cast<const v2f *>(fl + g[k]), fy = *reinterpret_cast<const v2f *>(fl + HW + g[k]);
            tp[k][0] = border_tap(xc, yc, fx.x, fy.x, gm, W, H);
            tp[k][1] = border_tap(xc + 1, yc, fx.y, fy.y, gm, W, H);
        }
        float v[3][2][2];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float *pl = src + c * HW;
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const float *q = pl + tp[k][e].off;
                    v[c][k][e] = blend(v2f{q[0], q[1]}, v2f{q[W], q[W + 1]}, tp[k][e]);
                }
        }
        if (L1) {
            const float *tgt = im1 + (long)b * 3 * HW;
            const float *oc = occ ? occ + (long)b * HW : nullptr;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                v2f tg[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) tg[c] = *reinterpret_cast<const v2f *>(tgt + c * HW + g[k]);
                v2f o = {1.f, 1.f};
                if (oc) o = *reinterpret_cast<const v2f *>(oc + g[k]);
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    float acc = 0.f;
#pragma unroll
                    for (int c = 0; c < 3; ++c) acc += fabsf(tg[c][e] - v[c][k][e]);
                    if (ok[k][e]) {
                        s += (double)(acc * o[e]);
                        so += (double)o[e];
                    }
                }
            }
        } else {
            float *ob = (float *)out_ + (long)b * 3 * HW;
#pragma unroll
            for (int k = 0; k < 2; ++k)
                if (ok[k][0])
#pragma unroll
                    for (int c = 0; c < 3; ++c) *reinterpret_cast<v2f *>(ob + c * HW + g[k]) = v2f{v[c][k][0], v[c][k][1]};
        }
    }
    if (L1) block_add2(s, so, (double *)out_);
}

// tile kernels for RGB / border calls; pad_mode | RCF_WARP_PER_PIXEL (a per-call choice: tests) keeps the per-pixel kernels, which
// also serve every other channel count and the zeros mode
inline bool warp_tile_applies(int C, int H, int W, int pad_mode) { return C == 3 && pad_mode == 0 && W >= 2 && H >= 2; }

inline WarpGeom warp_geom(int H, int W) {
    WarpGeom g;
    g.dx = (float)(W - 1);
    g.dy = (float)(H - 1);
    g.rx = 1.0f / g.dx;
    g.ry = 1.0f / g.dy;
    return g;
}

// sums[0] = sum |im-recon|*occ, sums[1] = sum occ, sums[2] = sum SSIM distance, sums[3] unused
__global__ void __launch_bounds__(256) photometric_kernel(const float *__restrict__ im, const float *__restrict__ rec,
                                                          const float *__restrict__ occ, double *__restrict__ sums,
                                                          int B, int C, int H, int W, int md) {
    const long HW = (long)H * W, total = (long)B * HW;
    const long step = (long)gridDim.x * blockDim.x;
    const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
    const int k = 2 * md + 1;
    const float inv = 1.f / (float)(k * k);
    double s1 = 0, so = 0, ss = 0, dummy = 0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += step) {
        const long b = i / HW, p = i - b * HW;
        const int py = (int)(p / W), px = (int)(p - (long)py * W);
        const float o = occ[i];
        so += (double)o;
        const bool interior = py >= md && py < H - md && px >= md && px < W - md;
        for (int c = 0; c < C; ++c) {
            const float *a = im + (b * C + c) * HW, *r = rec + (b * C + c) * HW;
            s1 += (double)(fabsf(a[p] - r[p]) * o);
            if (interior) {
                float sx = 0, sy = 0, sxx = 0, syy = 0, sxy = 0;
                for (int dy = -md; dy <= md; ++dy)
                    for (int dx = -md; dx <= md; ++dx) {
                        const long q = (long)(py + dy) * W + (px + dx);
                        const float oq = occ[b * HW + q];
                        const float xv = r[q] * oq, yv = a[q] * oq;   // SSIM(recon*occ, im*occ)
                        sx += xv; sy += yv; sxx += xv * xv; syy += yv * yv; sxy += xv * yv;
                    }
                const float mx = sx * inv, my = sy * inv;
                const float vx = sxx * inv - mx * mx, vy = syy * inv - my * my, cxy = sxy * inv - mx * my;
                const float n = (2.f * mx * my + C1) * (2.f * cxy + C2);
                const float d = (mx * mx + my * my + C1) * (vx + vy + C2);
                ss += (double)fminf(fmaxf((1.f - n / d) * 0.5f, 0.f), 1.f);
            }
        }
    }
    block_add2(s1, so, sums);
    block_add2(ss, dummy, sums + 2);
}

__global__ void photometric_final_kernel(const double *__restrict__ sums, float *__restrict__ out, float w_l1,
                                         float w_ssim, double n_l1, double n_ssim, double n_occ) {
    double loss = 0;
    if (w_l1 > 0) loss += (double)w_l1 * sums[0] / n_l1;
    if (w_ssim > 0) loss += (double)w_ssim * sums[2] / n_ssim;
    out[0] = (float)(loss / (sums[1] / n_occ));
}

}  // namespace

extern "C" int rcf_flow_warp_f32(const float *x, const float *flow, float *out, int B, int C, int H, int W,
                                 int pad_mode, void *stream) {
    const bool per_pixel = pad_mode & RCF_WARP_PER_PIXEL;
    pad_mode &= ~RCF_WARP_PER_PIXEL;
    if (!x || !flow || !out || B <= 0 || C <= 0 || H < 2 || W < 2 || (pad_mode != 0 && pad_mode != 1)) return RCF_EINVAL;
    if ((long)H * W >= (1L << 30)) return RCF_EINVAL;
    if (!per_pixel && warp_tile_applies(C, H, W, pad_mode) && W >= 256) {     // narrow images: the flat 512-pixel runs fill the lanes better
        const int tx = rcf_cdiv(W, TILE_W), ty = rcf_cdiv(H, 16);
        const long R = (long)B * rcf_cdiv((long)tx * ty, 8);
        const int Q = (int)(R < 1024 ? R : 1024);               // up to 8192 workgroups, each walking its share of the tiles
        hipLaunchKernelGGL((warp_rows_kernel<false, 4>), dim3((unsigned)(8 * Q)), dim3(256), 0, rcf_stream(stream),
                           (const float *)nullptr, x, flow, (const float *)nullptr, (void *)out, B, H, W, warp_geom(H, W), tx,
                           ty);
        RCF_LAUNCH_CHECK();
        return 0;
    }
    const int runs = rcf_cdiv((long)((H + 7) / 8) * W, 512);          // 512-pixel runs per band
    if ((long)8 * B * runs >= (1L << 31)) return RCF_EINVAL;
    hipLaunchKernelGGL(flow_warp_kernel, dim3((unsigned)(8 * B * runs)), dim3(256), 0, rcf_stream(stream), x, flow, out, B,
                       C, H, W, pad_mode, runs);
    RCF_LAUNCH_CHECK();
    return 0;
}

extern "C" int rcf_flow_warp_bwd_f32(const float *x, const float *flow, const float *dout, float *dx, float *dflow,
                                     int B, int C, int H, int W, int pad_mode, void *stream) {
    if (!x || !flow || !dout || B <= 0 || C <= 0 || H < 2 || W < 2 || (pad_mode != 0 && pad_mode != 1)) return RCF_EINVAL;
    hipLaunchKernelGGL(flow_warp_bwd_kernel, dim3(px_blocks((long)B * H * W)), dim3(256), 0, rcf_stream(stream), x,
                       flow, dout, dx, dflow, B, C, H, W, pad_mode);
    RCF_LAUNCH_CHECK();
    return 0;
}

extern "C" int rcf_occu_mask_backward_f32(const float *flow21, float *occ, float th, float *scratch, int B, int H,
                                          int W, void *stream) {
    if (!flow21 || !occ || !scratch || B <= 0 || H < 2 || W < 2) return RCF_EINVAL;
    hipStream_t st = rcf_stream(stream);
    const long n = (long)B * H * W;
    hipError_t e = hipMemsetAsync(scratch, 0, n * sizeof(float), st);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(splat_count_kernel, dim3(px_blocks(n)), dim3(256), 0, st, flow21, scratch, B, H, W);
    RCF_LAUNCH_CHECK();
    hipLaunchKernelGGL(occ_threshold_kernel, dim3(px_blocks(n)), dim3(256), 0, st, (const float *)scratch, occ, n, th);
    RCF_LAUNCH_CHECK();
    return 0;
}

extern "C" int rcf_occu_mask_bidirection_f32(const float *flow12, const float *flow21, float *occ, float scale,
                                             float bias, int B, int H, int W, void *stream) {
    if (!flow12 || !flow21 || !occ || B <= 0 || H < 2 || W < 2) return RCF_EINVAL;
    hipLaunchKernelGGL(occ_bidir_kernel, dim3(px_blocks((long)B * H * W)), dim3(256), 0, rcf_stream(stream), flow12,
                       flow21, occ, scale, bias, B, H, W);
    RCF_LAUNCH_CHECK();
    return 0;
}

extern "C" int rcf_warp_l1_residual_f32(const float *im1, const float *im2, const float *flow, const float *occ,
                                        double *out, int B, int C, int H, int W, int pad_mode, void *stream) {
    const bool per_pixel = pad_mode & RCF_WARP_PER_PIXEL;
    pad_mode &= ~RCF_WARP_PER_PIXEL;
    if (!im1 || !im2 || !flow || !out || B <= 0 || C <= 0 || H < 2 || W < 2) return RCF_EINVAL;
    hipStream_t st = rcf_stream(stream);
    hipError_t e = hipMemsetAsync(out, 0, 2 * sizeof(double), st);
    if (e != hipSuccess) return (int)e;
    if (!per_pixel && warp_tile_applies(C, H, W, pad_mode)) {
        const bool two = W % 2 == 0;                            // two pixels per lane, 8-byte stream loads
        const int tx = rcf_cdiv(W, two ? 128 : TILE_W), ty = rcf_cdiv(H, two ? 8 : 16);
        const long R = (long)B * rcf_cdiv((long)tx * ty, 8);
        // 4096 workgroups at most, and at least four tiles each: every workgroup ends in two fp64 atomics on one address
        const int Q = (int)(R < 4 ? 1 : (R / 4 < 512 ? R / 4 : 512));       // measured 64 / 128 / 256 / 512: 245 / 234 / 231 / 226 us
        hipLaunchKernelGGL((two ? warp_rows2_kernel<true> : warp_rows_kernel<true, 4>), dim3((unsigned)(8 * Q)), dim3(256), 0, st, im1, im2, flow, occ,
                           (void *)out, B, H, W, warp_geom(H, W), tx, ty);
        RCF_LAUNCH_CHECK();
        return 0;
    }
    // 8 bands x 256 workgroups (2048 in total: each ends in two fp64 atomics), fewer when there is less work
    const int runs = rcf_cdiv((long)((H + 7) / 8) * W, 512);          // 512-pixel runs per band
    const long R = (long)B * runs;
    const int Q = (int)(R < 256 ? R : 256);
    hipLaunchKernelGGL(warp_l1_kernel, dim3((unsigned)(8 * Q)), dim3(256), 0, st, im1, im2, flow, occ, out, B, C, H, W,
                       pad_mode, runs);
    RCF_LAUNCH_CHECK();
    return 0;
}

extern "C" int rcf_photometric_loss_f32(const float *im, const float *recon, const float *occ, float w_l1,
                                        float w_ssim, float *out, double *scratch, int B, int C, int H, int W,
                                        void *stream) {
    if (!im || !recon || !occ || !out || !scratch || B <= 0 || C <= 0 || H < 3 || W < 3) return RCF_EINVAL;
    hipStream_t st = rcf_stream(stream);
    hipError_t e = hipMemsetAsync(scratch, 0, 4 * sizeof(double), st);
    if (e != hipSuccess) return (int)e;
    long nb = ((long)B * H * W + 255) / 256;
    if (nb > 2048) nb = 2048;
    const int md = 1;
    hipLaunchKernelGGL(photometric_kernel, dim3((unsigned)nb), dim3(256), 0, st, im, recon, occ, scratch, B, C, H, W, md);
    RCF_LAUNCH_CHECK();
    hipLaunchKernelGGL(photometric_final_kernel, dim3(1), dim3(1), 0, st, (const double *)scratch, out, w_l1, w_ssim,
                       (double)B * C * H * W, (double)B * C * (H - 2 * md) * (W - 2 * md), (double)B * H * W);
    RCF_LAUNCH_CHECK();
    return 0;
}
