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
    if (!x || !flow || !out || B <= 0 || C <= 0 || H < 2 || W < 2 || (pad_mode != 0 && pad_mode != 1)) return RCF_EINVAL;
    if ((long)H * W >= (1L << 30)) return RCF_EINVAL;
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
    if (!im1 || !im2 || !flow || !out || B <= 0 || C <= 0 || H < 2 || W < 2) return RCF_EINVAL;
    hipStream_t st = rcf_stream(stream);
    hipError_t e = hipMemsetAsync(out, 0, 2 * sizeof(double), st);
    if (e != hipSuccess) return (int)e;
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
