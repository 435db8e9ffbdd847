// Pooling, bilinear resize and layout kernels (all HBM-bound; one float4 of channels per thread so a
// wavefront moves 1 KiB per access).  Reference call sites: MaxPool2d(3,2,1) models/resnet.py:577;
// mmseg `resize` (bilinear) models/decode_head.py:157-163, models/rcf_model.py:213-220,438-442;
// torch.cat models/decode_head.py:164; unflatten/flatten models/rcf_model.py:325.
// PyTorch index arithmetic is reproduced exactly.
#include "rcf_common.h"

namespace {

inline int ew_blocks(long total) {
    long b = (total + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

// ------------------------------------------------------------------------------------------ maxpool
template <typename T>
__global__ void __launch_bounds__(256) maxpool_fwd_kernel(const T *__restrict__ x, T *__restrict__ y,
                                                          uint8_t *__restrict__ am, int N, int H, int W, int C,
                                                          int Ho, int Wo) {
    const int CV = C / 4;
    const long total = (long)N * Ho * Wo * CV;
    const long step = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += step) {
        const int cv = (int)(i % CV);
        long t = i / CV;
        const int xo = (int)(t % Wo);
        t /= Wo;
        const int yo = (int)(t % Ho);
        const int n = (int)(t / Ho);
        f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        int bi[4] = {-1, -1, -1, -1};
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int yi = yo * 2 - 1 + r;
            if (yi < 0 || yi >= H) continue;
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const int xi = xo * 2 - 1 + s;
                if (xi < 0 || xi >= W) continue;
                const f32x4 v = ld4(x + (((long)n * H + yi) * W + xi) * C + cv * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    // torch max_pool2d: index starts at the first valid tap; later taps win on
                    // strict '>' (or NaN)
                    if (bi[e] < 0 || v[e] > best[e] || v[e] != v[e]) {
                        best[e] = v[e];
                        bi[e] = r * 3 + s;
                    }
                }
            }
        }
        st4(y + i * 4, best);
        uchar4 code = make_uchar4((uint8_t)bi[0], (uint8_t)bi[1], (uint8_t)bi[2], (uint8_t)bi[3]);
        *reinterpret_cast<uchar4 *>(am + i * 4) = code;
    }
}

template <typename T>
__global__ void __launch_bounds__(256) maxpool_bwd_kernel(const T *__restrict__ dy, const uint8_t *__restrict__ am,
                                                          T *__restrict__ dx, int N, int H, int W, int C, int Ho,
                                                          int Wo) {
    const int CV = C / 4;
    const long total = (long)N * H * W * CV;
    const long step = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += step) {
        const int cv = (int)(i % CV);
        long t = i / CV;
        const int xi = (int)(t % W);
        t /= W;
        const int yi = (int)(t % H);
        const int n = (int)(t / H);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int ty = yi + 1 - r;
            if (ty < 0 || (ty & 1)) continue;
            const int yo = ty >> 1;
            if (yo >= Ho) continue;
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const int tx = xi + 1 - s;
                if (tx < 0 || (tx & 1)) continue;
                const int xo = tx >> 1;
                if (xo >= Wo) continue;
                const long o = ((((long)n * Ho + yo) * Wo + xo) * CV + cv) * 4;
                const uchar4 code = *reinterpret_cast<const uchar4 *>(am + o);
                const f32x4 g = ld4(dy + o);
                const int want = r * 3 + s;
                if (code.x == want) acc[0] += g[0];
                if (code.y == want) acc[1] += g[1];
                if (code.z == want) acc[2] += g[2];
                if (code.w == want) acc[3] += g[3];
            }
        }
        st4(dx + i * 4, acc);
    }
}

// ------------------------------------------------------------------------------------------- resize
// PyTorch area_pixel_compute_source_index (UpSample.h): align_corners ? scale*dst
//   : max(scale*(dst+0.5)-0.5, 0); scale = align ? (in-1)/(out-1) : in/out.
__device__ __forceinline__ void src_index(int dst, float scale, int align, int in_size, int &i0, int &i1, float &l1) {
    float s = align ? scale * dst : fmaxf(scale * (dst + 0.5f) - 0.5f, 0.f);
    i0 = (int)s;
    if (i0 > in_size - 1) i0 = in_size - 1;
    i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
    l1 = s - i0;
}
inline float host_scale(int in, int out, int align) {
    if (align) return out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f;
    return (float)in / (float)out;
}

// pixel q of the border frame of thickness t of an H x W image (top strip, bottom strip, left, right; row-major each)
__device__ __forceinline__ void frame_yx(int q, int H, int W, int t, int &y, int &x) {
    const int strip = t * W;
    if (q < 2 * strip) {
        const int bottom = q >= strip;
        q -= bottom ? strip : 0;
        y = q / W + (bottom ? H - t : 0);
        x = q % W;
    } else {
        q -= 2 * strip;
        const int side = t * (H - 2 * t);
        const int right = q >= side;
        q -= right ? side : 0;
        y = t + q / t;
        x = q % t + (right ? W - t : 0);
    }
}
__device__ __forceinline__ bool in_frame(int y, int x, int H, int W, int t) {
    return y < t || y >= H - t || x < t || x >= W - t;
}

// frame > 0: only the output pixels within `frame` of the border are written
template <typename T>
__global__ void __launch_bounds__(256) resize_nhwc_fwd_kernel(const T *__restrict__ x, int x_pitch,
                                                              T *__restrict__ y, int y_pitch, int N, int Hi,
                                                              int Wi, int Ho, int Wo, int C, int align, float sh,
                                                              float sw, int frame) {
    const int CV = C / 4;
    const int per_img = frame > 0 ? 2 * frame * Wo + 2 * frame * (Ho - 2 * frame) : Ho * Wo;
    const long total = (long)N * per_img * CV;
    const long step = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += step) {
        const int cv = (int)(i % CV);
        long t = i / CV;
        const int q = (int)(t % per_img);
        const int n = (int)(t / per_img);
        int yo, xo;
        if (frame > 0) frame_yx(q, Ho, Wo, frame, yo, xo);
        else { yo = q / Wo; xo = q - yo * Wo; }
        int y0, y1, x0, x1;
        float ly, lx;
        src_index(yo, sh, align, Hi, y0, y1, ly);
        src_index(xo, sw, align, Wi, x0, x1, lx);
        const T *base = x + (long)n * Hi * Wi * x_pitch + cv * 4;
        const f32x4 v00 = ld4(base + ((long)y0 * Wi + x0) * x_pitch);
        const f32x4 v01 = ld4(base + ((long)y0 * Wi + x1) * x_pitch);
        const f32x4 v10 = ld4(base + ((long)y1 * Wi + x0) * x_pitch);
        const f32x4 v11 = ld4(base + ((long)y1 * Wi + x1) * x_pitch);
        const float hy = 1.f - ly, hx = 1.f - lx;
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = hy * (hx * v00[e] + lx * v01[e]) + ly * (hx * v10[e] + lx * v11[e]);
        st4(y + (((long)n * Ho + yo) * Wo + xo) * y_pitch + cv * 4, o);
    }
}


// weights of output index o onto input index `in_idx`
__device__ __forceinline__ float tap_weight(int o, int in_idx, float scale, int align, int in_size) {
    int i0, i1;
    float l1;
    src_index(o, scale, align, in_size, i0, i1, l1);
    float w = 0.f;
    if (i0 == in_idx) w += 1.f - l1;
    if (i1 == in_idx) w += l1;
    return w;
}
__device__ __forceinline__ void cand_range(int in_idx, float scale, int align, int out_size, int &lo, int &hi) {
    if (scale <= 0.f) { lo = 0; hi = out_size - 1; return; }
    const float inv = 1.f / scale;
    float a, b;
    if (align) { a = (in_idx - 1) * inv; b = (in_idx + 1) * inv; }
    else { a = (in_idx - 0.5f) * inv - 0.5f; b = (in_idx + 1.5f) * inv - 0.5f; }
    lo = (int)floorf(a) - 1;
    hi = (int)ceilf(b) + 1;
    if (lo < 0) lo = 0;
    if (hi > out_size - 1) hi = out_size - 1;
}

// frame > 0: dy is taken as zero outside the border frame of that thickness (and not read there)
template <typename T>
__global__ void __launch_bounds__(256) resize_nhwc_bwd_kernel(const T *__restrict__ dy, int dy_pitch,
                                                              T *__restrict__ dx, int dx_pitch, int beta, int N,
                                                              int Hi, int Wi, int Ho, int Wo, int C, int align,
                                                              float sh, float sw, int frame, int tc) {
    // tc > 0 (frame > 0 and beta != 0): only the input pixels within tc of the border can receive anything from the
    // frame -- the walk covers that input frame alone; the interior keeps what it holds
    const int CV = C / 4;
    const long per_img = tc > 0 ? 2L * tc * Wi + 2L * tc * (Hi - 2 * tc) : (long)Hi * Wi;
    const long total = (long)N * per_img * CV;
    const long step = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += step) {
        const int cv = (int)(i % CV);
        long t = i / CV;
        const int q = (int)(t % per_img);
        const int n = (int)(t / per_img);
        int xi, yi;
        if (tc > 0) {
            frame_yx(q, Hi, Wi, tc, yi, xi);
        } else {
            yi = q / Wi;
            xi = q - yi * Wi;
        }
        int ylo, yhi, xlo, xhi;
        cand_range(yi, sh, align, Ho, ylo, yhi);
        cand_range(xi, sw, align, Wo, xlo, xhi);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (frame > 0 && ylo >= frame && yhi < Ho - frame && xlo >= frame && xhi < Wo - frame) {
            // no candidate lies on the frame: nothing reaches this pixel (most of the image)
            if (!beta) st4(dx + (((long)n * Hi + yi) * Wi + xi) * dx_pitch + cv * 4, acc);
            continue;
        }
        constexpr int MAXC = 6;                 // candidates per axis handled with weights computed once (2x: 4)
        if (yhi - ylo < MAXC && xhi - xlo < MAXC) {
            // the column weights are the same for every candidate row: computed once, not once per (row, column)
            float wxv[MAXC];
#pragma unroll
            for (int k = 0; k < MAXC; ++k) wxv[k] = xlo + k <= xhi ? tap_weight(xlo + k, xi, sw, align, Wi) : 0.f;
            for (int yo = ylo; yo <= yhi; ++yo) {
                const float wy = tap_weight(yo, yi, sh, align, Hi);
                if (wy == 0.f) continue;
                const T *row = dy + (((long)n * Ho + yo) * Wo + xlo) * dy_pitch + cv * 4;
#pragma unroll
                for (int k = 0; k < MAXC; ++k) {
                    if (xlo + k > xhi || wxv[k] == 0.f) continue;
                    if (frame > 0 && !in_frame(yo, xlo + k, Ho, Wo, frame)) continue;
                    const f32x4 g = ld4(row + (long)k * dy_pitch);
                    acc += g * (wy * wxv[k]);
                }
            }
        } else {
            for (int yo = ylo; yo <= yhi; ++yo) {
                const float wy = tap_weight(yo, yi, sh, align, Hi);
                if (wy == 0.f) continue;
                for (int xo = xlo; xo <= xhi; ++xo) {
                    const float wx = tap_weight(xo, xi, sw, align, Wi);
                    if (wx == 0.f) continue;
                    if (frame > 0 && !in_frame(yo, xo, Ho, Wo, frame)) continue;
                    const f32x4 g = ld4(dy + (((long)n * Ho + yo) * Wo + xo) * dy_pitch + cv * 4);
                    acc += g * (wy * wx);
                }
            }
        }
        T *dst = dx + (((long)n * Hi + yi) * Wi + xi) * dx_pitch + cv * 4;
        st4(dst, beta ? (ld4(dst) + acc) : acc);
    }
}

// Whole-tensor forms (frame == 0) of the two kernels above, which is what the step runs on its large tensors
// ([16,60,107,256] <-> [16,120,214,256]): blockIdx.y = (image, row), so no thread divides a 64-bit index three times,
// and V channels per thread (8 = one 16-byte access for bf16).  rocprof before: fwd 221 us, bwd 501 us on that tensor
// in bf16 (1.2 / 0.5 TB/s of the bytes moved).  Same arithmetic, same order: identical results.
template <typename T, int V>
__global__ void __launch_bounds__(256) resize_rows_fwd_kernel(const T *__restrict__ x, int x_pitch, T *__restrict__ y,
                                                              int y_pitch, int Hi, int Wi, int Ho, int Wo, int C, int align,
                                                              float sh, float sw, int frame) {
    const int CV = C / V;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int xo = idx / CV, cv = idx - xo * CV;
    if (xo >= Wo) return;
    const int n = blockIdx.y / Ho, yo = blockIdx.y - n * Ho;
    if (frame > 0 && !in_frame(yo, xo, Ho, Wo, frame)) return;      // only the border frame is written
    int y0, y1, x0, x1;
    float ly, lx;
    src_index(yo, sh, align, Hi, y0, y1, ly);
    src_index(xo, sw, align, Wi, x0, x1, lx);
    const T *base = x + (long)n * Hi * Wi * x_pitch + cv * V;
    const fvec<V> v00 = ldv<T, V>(base + ((long)y0 * Wi + x0) * x_pitch), v01 = ldv<T, V>(base + ((long)y0 * Wi + x1) * x_pitch);
    const fvec<V> v10 = ldv<T, V>(base + ((long)y1 * Wi + x0) * x_pitch), v11 = ldv<T, V>(base + ((long)y1 * Wi + x1) * x_pitch);
    const float hy = 1.f - ly, hx = 1.f - lx;
    fvec<V> o;
#pragma unroll
    for (int h = 0; h < V / 4; ++h)
#pragma unroll
        for (int e = 0; e < 4; ++e)
            o.q[h][e] = hy * (hx * v00.q[h][e] + lx * v01.q[h][e]) + ly * (hx * v10.q[h][e] + lx * v11.q[h][e]);
    stv<T, V>(y + (((long)n * Ho + yo) * Wo + xo) * y_pitch + cv * V, o);
}

// Exact 2x up-sampling without align_corners (the decode heads' 60x107 -> 120x214: every bilinear resize of the step): a
// thread produces the 2 x 2 outputs of ONE source pixel from the 3 x 3 source pixels around it -- 9 loads for 4 outputs where
// resize_rows_fwd_kernel issues 16 (its 230 us per 420 MB of output were L2 read bandwidth: 7.3 TB/s of 16-byte gathers).
// Same taps, same weights (src_index itself), same expression tree: hy (hx a + lx b) + ly (hx c + lx d) -- the horizontal blends
// of the two source rows first, as there -- so the outputs are bit-identical.
// frame > 0 (round 6): only the output pixels within `frame` of the border are written, and the grid holds only the source pixels
// that have such an output -- the border frame of thickness ts = (frame + 1) / 2 of the source, enumerated by frame_yx, grid
// (its pixels x channel vectors / 256, images).  The whole-tensor grid of resize_rows_fwd_kernel returned at once in 70 % of its
// 8 x 10^5 workgroups on decode_head2's border band and spent 16 loads per output on the rest: 523 -> ~300 us.
template <typename T, int V>
__global__ void __launch_bounds__(256) resize2x_fwd_kernel(const T *__restrict__ x, int x_pitch, T *__restrict__ y,
                                                           int y_pitch, int Hi, int Wi, int C, int frame = 0, int ts = 0) {
    const int CV = C / V;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    int n, i, j;
    const int q = idx / CV, cv = idx - q * CV;
    if (frame > 0) {
        if (q >= 2 * ts * Wi + 2 * ts * (Hi - 2 * ts)) return;
        n = blockIdx.y;
        frame_yx(q, Hi, Wi, ts, i, j);
    } else {
        j = q;
        if (j >= Wi) return;
        n = blockIdx.y / Hi;
        i = blockIdx.y - n * Hi;                                            // block-uniform
    }
    const int Ho = 2 * Hi, Wo = 2 * Wi;
    const int rr[3] = {max(i - 1, 0), i, min(i + 1, Hi - 1)}, cc[3] = {max(j - 1, 0), j, min(j + 1, Wi - 1)};
    const T *base = x + (long)n * Hi * Wi * x_pitch + cv * V;
    fvec<V> v[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) v[r][c] = ldv<T, V>(base + ((long)rr[r] * Wi + cc[c]) * x_pitch);
    // output column 2j + b takes source columns (x0, x1) = slots (kx, kx + 1) of cc: slot 0 unless the left border clamps
    fvec<V> h[3][2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        int x0, x1;
        float lx;
        src_index(2 * j + b, 0.5f, 0, Wi, x0, x1, lx);
        const float hx = 1.f - lx;
        const bool first = b == 0 && j > 0;                                   // (j-1, j); otherwise (j, j+1) -- at j == 0, b == 0: (0, 1)
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int q = 0; q < V / 4; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float a = first ? v[r][0].q[q][e] : v[r][1].q[q][e];
                    const float bb = first ? v[r][1].q[q][e] : v[r][2].q[q][e];
                    h[r][b].q[q][e] = hx * a + lx * bb;
                }
    }
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        int y0, y1;
        float ly;
        src_index(2 * i + a, 0.5f, 0, Hi, y0, y1, ly);
        const float hy = 1.f - ly;
        const bool first = a == 0 && i > 0;                                   // rows (i-1, i); otherwise (i, i+1) -- at i == 0, a == 0: (0, 1)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            fvec<V> o;
#pragma unroll
            for (int q = 0; q < V / 4; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float t = first ? h[0][b].q[q][e] : h[1][b].q[q][e];
                    const float u = first ? h[1][b].q[q][e] : h[2][b].q[q][e];
                    o.q[q][e] = hy * t + ly * u;
                }
            if (frame > 0 && !in_frame(2 * i + a, 2 * j + b, Ho, Wo, frame)) continue;
            stv<T, V>(y + (((long)n * Ho + 2 * i + a) * Wo + 2 * j + b) * y_pitch + cv * V, o);
        }
    }
}

// Backward of the exact 2x up-sampling: a thread gathers for a 2 x 2 block of input pixels from the 6 x 6 output pixels that
// can reach it, one output row at a time -- 36 loads for 4 inputs where resize_rows_bwd_kernel issues 16 per input.  Every
// input accumulates the same terms g * (wy * wx) (tap_weight itself, zero weights skipped) in the same order (output rows
// upwards, columns upwards): bit-identical.
// frame > 0 (round 6; beta = 1 only): dy counts as zero (and is not read) off the border frame, and the grid holds only the 2 x 2
// input blocks within tb blocks of the border (the others see no frame pixel and, accumulating, have nothing to write): the same
// terms in the same order as resize_rows_bwd_kernel's frame form, which gathered 16 taps per input pixel (686 us on
// decode_head2's band in fp32: 4 GB of 16-byte L2 reads).
template <typename T, int V>
__global__ void __launch_bounds__(256) resize2x_bwd_kernel(const T *__restrict__ dy, int dy_pitch, T *__restrict__ dx,
                                                           int dx_pitch, int beta, int Hi, int Wi, int C, int frame = 0, int tb = 0) {
    const int CV = C / V, Wh = (Wi + 1) >> 1, Hh = (Hi + 1) >> 1;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    int n, i2, j2;
    const int q = idx / CV, cv = idx - q * CV;
    if (frame > 0) {
        if (q >= 2 * tb * Wh + 2 * tb * (Hh - 2 * tb)) return;
        n = blockIdx.y;
        frame_yx(q, Hh, Wh, tb, i2, j2);
    } else {
        j2 = q;
        if (j2 >= Wh) return;
        n = blockIdx.y / Hh;
        i2 = blockIdx.y - n * Hh;                                           // block-uniform
    }
    const int Ho = 2 * Hi, Wo = 2 * Wi;
    const int i0 = 2 * i2, j0 = 2 * j2;
    fvec<V> acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int h = 0; h < V / 4; ++h) acc[a][b].q[h] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int xo0 = 2 * j0 - 1;                                             // first candidate column; candidates xo0 .. xo0 + 5
    float wx[2][6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const int xo = xo0 + k;
        const bool ok = xo >= 0 && xo < Wo;
        wx[0][k] = ok ? tap_weight(xo, j0, 0.5f, 0, Wi) : 0.f;
        wx[1][k] = ok && j0 + 1 < Wi ? tap_weight(xo, j0 + 1, 0.5f, 0, Wi) : 0.f;
    }
    for (int yo = max(2 * i0 - 1, 0); yo <= min(2 * i0 + 4, Ho - 1); ++yo) {
        const float wy0 = tap_weight(yo, i0, 0.5f, 0, Hi);
        const float wy1 = i0 + 1 < Hi ? tap_weight(yo, i0 + 1, 0.5f, 0, Hi) : 0.f;
        if (wy0 == 0.f && wy1 == 0.f) continue;
        const T *row = dy + (((long)n * Ho + yo) * Wo + xo0) * dy_pitch + cv * V;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            if (wx[0][k] == 0.f && wx[1][k] == 0.f) continue;               // also: column out of range
            if (frame > 0 && !in_frame(yo, xo0 + k, Ho, Wo, frame)) continue;
            const fvec<V> g = ldv<T, V>(row + (long)k * dy_pitch);
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const float wy = a ? wy1 : wy0;
                if (wy == 0.f) continue;
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    if (wx[b][k] == 0.f) continue;
                    const float w = wy * wx[b][k];
#pragma unroll
                    for (int h = 0; h < V / 4; ++h) acc[a][b].q[h] += g.q[h] * w;
                }
            }
        }
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            if (i0 + a >= Hi || j0 + b >= Wi) continue;
            T *dst = dx + (((long)n * Hi + i0 + a) * Wi + j0 + b) * dx_pitch + cv * V;
            if (beta) {
                const fvec<V> old = ldv<T, V>(dst);
#pragma unroll
                for (int h = 0; h < V / 4; ++h) acc[a][b].q[h] = old.q[h] + acc[a][b].q[h];
            }
            stv<T, V>(dst, acc[a][b]);
        }
}

template <typename T, int V>
__global__ void __launch_bounds__(256) resize_rows_bwd_kernel(const T *__restrict__ dy, int dy_pitch, T *__restrict__ dx,
                                                              int dx_pitch, int beta, int Hi, int Wi, int Ho, int Wo, int C,
                                                              int align, float sh, float sw, int frame) {
    const int CV = C / V;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int xi = idx / CV, cv = idx - xi * CV;
    if (xi >= Wi) return;
    const int n = blockIdx.y / Hi, yi = blockIdx.y - n * Hi;
    int ylo, yhi, xlo, xhi;
    cand_range(yi, sh, align, Ho, ylo, yhi);
    cand_range(xi, sw, align, Wo, xlo, xhi);
    fvec<V> acc;
#pragma unroll
    for (int h = 0; h < V / 4; ++h) acc.q[h] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (frame > 0 && ylo >= frame && yhi < Ho - frame && xlo >= frame && xhi < Wo - frame) {
        // frame > 0: dy counts as zero off the border frame; no candidate of this pixel lies on it (most of the image)
        if (!beta) stv<T, V>(dx + (((long)n * Hi + yi) * Wi + xi) * dx_pitch + cv * V, acc);
        return;
    }
    constexpr int MAXC = 6;
    if (yhi - ylo < MAXC && xhi - xlo < MAXC) {
        float wxv[MAXC];
#pragma unroll
        for (int k = 0; k < MAXC; ++k) wxv[k] = xlo + k <= xhi ? tap_weight(xlo + k, xi, sw, align, Wi) : 0.f;
        for (int yo = ylo; yo <= yhi; ++yo) {
            const float wy = tap_weight(yo, yi, sh, align, Hi);
            if (wy == 0.f) continue;
            const T *row = dy + (((long)n * Ho + yo) * Wo + xlo) * dy_pitch + cv * V;
#pragma unroll
            for (int k = 0; k < MAXC; ++k) {
                if (xlo + k > xhi || wxv[k] == 0.f) continue;
                if (frame > 0 && !in_frame(yo, xlo + k, Ho, Wo, frame)) continue;
                const fvec<V> g = ldv<T, V>(row + (long)k * dy_pitch);
                const float w = wy * wxv[k];
#pragma unroll
                for (int h = 0; h < V / 4; ++h) acc.q[h] += g.q[h] * w;
            }
        }
    } else {
        for (int yo = ylo; yo <= yhi; ++yo) {
            const float wy = tap_weight(yo, yi, sh, align, Hi);
            if (wy == 0.f) continue;
            for (int xo = xlo; xo <= xhi; ++xo) {
                const float wx = tap_weight(xo, xi, sw, align, Wi);
                if (wx == 0.f) continue;
                if (frame > 0 && !in_frame(yo, xo, Ho, Wo, frame)) continue;
                const fvec<V> g = ldv<T, V>(dy + (((long)n * Ho + yo) * Wo + xo) * dy_pitch + cv * V);
                const float w = wy * wx;
#pragma unroll
                for (int h = 0; h < V / 4; ++h) acc.q[h] += g.q[h] * w;
            }
        }
    }
    T *dst = dx + (((long)n * Hi + yi) * Wi + xi) * dx_pitch + cv * V;
    if (beta) {
        const fvec<V> old = ldv<T, V>(dst);
#pragma unroll
        for (int h = 0; h < V / 4; ++h) acc.q[h] = old.q[h] + acc.q[h];
    }
    stv<T, V>(dst, acc);
}

__global__ void __launch_bounds__(256) resize_nchw_kernel(const float *__restrict__ x, float *__restrict__ y,
                                                          int planes, int Hi, int Wi, int Ho, int Wo, int align,
                                                          float sh, float sw) {
    const long total = (long)planes * Ho * Wo;
    const long step = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += step) {
        const int xo = (int)(i % Wo);
        long t = i / Wo;
        const int yo = (int)(t % Ho);
        const long pl = t / Ho;
        int y0, y1, x0, x1;
        float ly, lx;
        src_index(yo, sh, align, Hi, y0, y1, ly);
        src_index(xo, sw, align, Wi, x0, x1, lx);
        const float *b = x + pl * Hi * Wi;
        const float hy = 1.f - ly, hx = 1.f - lx;
        y[i] = hy * (hx * b[(long)y0 * Wi + x0] + lx * b[(long)y0 * Wi + x1]) +
               ly * (hx * b[(long)y1 * Wi + x0] + lx * b[(long)y1 * Wi + x1]);
    }
}

// ------------------------------------------------------------------------------------------- layout
__global__ void __launch_bounds__(256) nchw_to_nhwc_kernel(const float *__restrict__ x, float *__restrict__ y, int N,
                                                           int C, long HW, int Cpad) {
    const long total = (long)N * HW;
    const long step = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += step) {
        const long n = i / HW, p = i - n * HW;
        const float *src = x + n * C * HW + p;
        float *dst = y + i * Cpad;
        for (int c = 0; c < Cpad; c += 4) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (c + e) < C ? src[(long)(c + e) * HW] : 0.f;
            *reinterpret_cast<f32x4 *>(dst + c) = v;
        }
    }
}

__global__ void __launch_bounds__(256) nhwc_to_nchw_kernel(const float *__restrict__ x, int x_pitch,
                                                           float *__restrict__ y, int N, int C, long HW) {
    const long total = (long)N * C * HW;
    const long step = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += step) {
        const long p = i % HW;
        const long t = i / HW;
        const int c = (int)(t % C);
        const long n = t / C;
        y[i] = x[(n * HW + p) * x_pitch + c];
    }
}

// ST -> DT: the same kernel is the fp32 <-> bf16 cast
template <typename ST, typename DT>
__global__ void __launch_bounds__(256) copy2d_kernel(const ST *__restrict__ src, long spitch,
                                                     DT *__restrict__ dst, long dpitch, long rows, int C, int beta) {
    const int CV = C / 4;
    const long total = rows * CV;
    const long step = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += step) {
        const long r = i / CV;
        const int c4 = (int)(i - r * CV) * 4;
        const f32x4 v = ld4(src + r * spitch + c4);
        DT *d = dst + r * dpitch + c4;
        st4(d, beta ? (ld4(d) + v) : v);
    }
}

// n0 x n1 copies in one launch (blockIdx.y = i0 * n1 + i1): copy (i0, i1) starts at src + i0*sb0 + i1*sb1 and
// dst + i0*db0 + i1*db1 (element offsets, may be negative)
template <typename T>
__global__ void __launch_bounds__(256) copy2d_batched_kernel(const T *__restrict__ src, long spitch, long sb0, long sb1,
                                                             T *__restrict__ dst, long dpitch, long db0, long db1,
                                                             long rows, int C, int beta, int n1) {
    const int i0 = blockIdx.y / n1, i1 = blockIdx.y - i0 * n1;
    src += i0 * sb0 + i1 * sb1;
    dst += i0 * db0 + i1 * db1;
    const int CV = C / 4;
    const long total = rows * CV;
    const long step = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += step) {
        const long r = i / CV;
        const int c4 = (int)(i - r * CV) * 4;
        const f32x4 v = ld4(src + r * spitch + c4);
        T *d = dst + r * dpitch + c4;
        st4(d, beta ? (ld4(d) + v) : v);
    }
}

// inside[p] = src[p] for pixels in the rectangle, 0 elsewhere; outside[p] = the complement (either may be null)
template <typename T>
__global__ void __launch_bounds__(256) split_rect_kernel(const T *__restrict__ src, T *__restrict__ inside,
                                                         T *__restrict__ outside, long pixels, int H, int W, int C,
                                                         int y0, int x0, int h, int w) {
    const int CV = C / 4;
    const long total = pixels * CV;
    const long step = (long)gridDim.x * blockDim.x;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += step) {
        const long p = i / CV;
        const int x = (int)(p % W), y = (int)((p / W) % H);
        const bool in = y >= y0 && y < y0 + h && x >= x0 && x < x0 + w;
        const f32x4 v = ld4(src + i * 4);
        if (inside) st4(inside + i * 4, in ? v : z);
        if (outside) st4(outside + i * 4, in ? z : v);
    }
}

}  // namespace

extern "C" int rcf_split_rect_mp(const void *src, void *inside, void *outside, int dt, int N, int H, int W, int C, int y0,
                                 int x0, int h, int w, void *stream) {
    if (!src || (!inside && !outside) || N <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 4) return RCF_EINVAL;
    if (y0 < 0 || x0 < 0 || h <= 0 || w <= 0 || y0 + h > H || x0 + w > W) return RCF_EINVAL;
    const long pixels = (long)N * H * W;
#define RCF_CALL(T)                                                                                                    \
    hipLaunchKernelGGL(split_rect_kernel<T>, dim3(ew_blocks(pixels * (C / 4))), dim3(256), 0, rcf_stream(stream),     \
                       (const T *)src, (T *)inside, (T *)outside, pixels, H, W, C, y0, x0, h, w)
    RCF_DISPATCH1(dt, RCF_CALL);
#undef RCF_CALL
    RCF_LAUNCH_CHECK();
    return 0;
}
extern "C" int rcf_split_rect_f32(const float *src, float *inside, float *outside, int N, int H, int W, int C, int y0,
                                  int x0, int h, int w, void *stream) {
    return rcf_split_rect_mp(src, inside, outside, RCF_F32, N, H, W, C, y0, x0, h, w, stream);
}

extern "C" int rcf_maxpool3x3s2_fwd_mp(const void *x, void *y, int dt, uint8_t *argmax, int N, int H, int W, int C, int Ho,
                                       int Wo, void *stream) {
    if (!x || !y || !argmax || C % 4 || Ho != (H + 2 - 3) / 2 + 1 || Wo != (W + 2 - 3) / 2 + 1) return RCF_EINVAL;
#define RCF_CALL(T)                                                                                               \
    hipLaunchKernelGGL(maxpool_fwd_kernel<T>, dim3(ew_blocks((long)N * Ho * Wo * (C / 4))), dim3(256), 0,        \
                       rcf_stream(stream), (const T *)x, (T *)y, argmax, N, H, W, C, Ho, Wo)
    RCF_DISPATCH1(dt, RCF_CALL);
#undef RCF_CALL
    RCF_LAUNCH_CHECK();
    return 0;
}
extern "C" int rcf_maxpool3x3s2_fwd_f32(const float *x, float *y, uint8_t *argmax, int N, int H, int W, int C, int Ho,
                                        int Wo, void *stream) {
    return rcf_maxpool3x3s2_fwd_mp(x, y, RCF_F32, argmax, N, H, W, C, Ho, Wo, stream);
}

extern "C" int rcf_maxpool3x3s2_bwd_mp(const void *dy, const uint8_t *argmax, void *dx, int dt, int N, int H, int W, int C,
                                       int Ho, int Wo, void *stream) {
    if (!dy || !dx || !argmax || C % 4) return RCF_EINVAL;
#define RCF_CALL(T)                                                                                                       \
    hipLaunchKernelGGL(maxpool_bwd_kernel<T>, dim3(ew_blocks((long)N * H * W * (C / 4))), dim3(256), 0, rcf_stream(stream), \
                       (const T *)dy, argmax, (T *)dx, N, H, W, C, Ho, Wo)
    RCF_DISPATCH1(dt, RCF_CALL);
#undef RCF_CALL
    RCF_LAUNCH_CHECK();
    return 0;
}
extern "C" int rcf_maxpool3x3s2_bwd_f32(const float *dy, const uint8_t *argmax, float *dx, int N, int H, int W, int C,
                                        int Ho, int Wo, void *stream) {
    return rcf_maxpool3x3s2_bwd_mp(dy, argmax, dx, RCF_F32, N, H, W, C, Ho, Wo, stream);
}

/* frame == 0: the whole tensor; frame == -1: the whole tensor on the general kernel even where the exact-2x form applies (the two
 * are bit-identical: tests) */
extern "C" int rcf_resize_bilinear_nhwc_fwd_mp(const void *x, int x_pitch, void *y, int y_pitch, int dt, int N, int Hi,
                                               int Wi, int Ho, int Wo, int C, int align_corners, int frame, void *stream) {
    if (!x || !y || C % 4 || x_pitch % 4 || y_pitch % 4 || Hi <= 0 || Wi <= 0 || Ho <= 0 || Wo <= 0) return RCF_EINVAL;
    const bool general = frame == -1;
    if (general) frame = 0;
    if (frame < 0 || (frame > 0 && (2 * frame >= Ho || 2 * frame >= Wo))) return RCF_EINVAL;
    const long px = frame > 0 ? (long)N * (2L * frame * Wo + 2L * frame * (Ho - 2 * frame)) : (long)N * Ho * Wo;
    if (!general && frame == 0 && !align_corners && Ho == 2 * Hi && Wo == 2 * Wi && Hi >= 2 && Wi >= 2 && (long)N * Hi <= 65535 &&
        (long)Wi * (C / 4) < (1L << 30)) {
        if (dt == RCF_BF16 && C % 8 == 0 && x_pitch % 8 == 0 && y_pitch % 8 == 0) {
            hipLaunchKernelGGL((resize2x_fwd_kernel<bf16_t, 8>), dim3(rcf_cdiv((long)Wi * (C / 8), 256), N * Hi), dim3(256), 0,
                               rcf_stream(stream), (const bf16_t *)x, x_pitch, (bf16_t *)y, y_pitch, Hi, Wi, C, 0, 0);
        } else {
#define RCF_CALL(T)                                                                                                       \
    hipLaunchKernelGGL((resize2x_fwd_kernel<T, 4>), dim3(rcf_cdiv((long)Wi * (C / 4), 256), N * Hi), dim3(256), 0,         \
                       rcf_stream(stream), (const T *)x, x_pitch, (T *)y, y_pitch, Hi, Wi, C, 0, 0)
            RCF_DISPATCH1(dt, RCF_CALL);
#undef RCF_CALL
        }
        RCF_LAUNCH_CHECK();
        return 0;
    }
    // the border frame of an exact 2x up-sampling (decode_head2's commuted conv): the 2x kernel over the source pixels that reach it
    const int ts = (frame + 1) / 2;
    if (!general && frame > 0 && !align_corners && Ho == 2 * Hi && Wo == 2 * Wi && 2 * ts < Hi && 2 * ts < Wi && N <= 65535 &&
        (long)Wi * Hi * (C / 4) < (1L << 30)) {
        const long Qs = 2L * ts * Wi + 2L * ts * (Hi - 2 * ts);
        if (dt == RCF_BF16 && C % 8 == 0 && x_pitch % 8 == 0 && y_pitch % 8 == 0) {
            hipLaunchKernelGGL((resize2x_fwd_kernel<bf16_t, 8>), dim3(rcf_cdiv(Qs * (C / 8), 256), N), dim3(256), 0,
                               rcf_stream(stream), (const bf16_t *)x, x_pitch, (bf16_t *)y, y_pitch, Hi, Wi, C, frame, ts);
        } else {
#define RCF_CALL(T)                                                                                                       \
    hipLaunchKernelGGL((resize2x_fwd_kernel<T, 4>), dim3(rcf_cdiv(Qs * (C / 4), 256), N), dim3(256), 0,                     \
                       rcf_stream(stream), (const T *)x, x_pitch, (T *)y, y_pitch, Hi, Wi, C, frame, ts)
            RCF_DISPATCH1(dt, RCF_CALL);
#undef RCF_CALL
        }
        RCF_LAUNCH_CHECK();
        return 0;
    }
    if ((long)N * Ho <= 65535 && (long)Wo * (C / 4) < (1L << 30)) {
        const float sh = host_scale(Hi, Ho, align_corners), sw = host_scale(Wi, Wo, align_corners);
        if (dt == RCF_BF16 && C % 8 == 0 && x_pitch % 8 == 0 && y_pitch % 8 == 0) {
            hipLaunchKernelGGL((resize_rows_fwd_kernel<bf16_t, 8>), dim3(rcf_cdiv((long)Wo * (C / 8), 256), N * Ho), dim3(256), 0,
                               rcf_stream(stream), (const bf16_t *)x, x_pitch, (bf16_t *)y, y_pitch, Hi, Wi, Ho, Wo, C,
                               align_corners, sh, sw, frame);
        } else {
#define RCF_CALL(T)                                                                                                       \
    hipLaunchKernelGGL((resize_rows_fwd_kernel<T, 4>), dim3(rcf_cdiv((long)Wo * (C / 4), 256), N * Ho), dim3(256), 0,      \
                       rcf_stream(stream), (const T *)x, x_pitch, (T *)y, y_pitch, Hi, Wi, Ho, Wo, C, align_corners, sh, sw, frame)
            RCF_DISPATCH1(dt, RCF_CALL);
#undef RCF_CALL
        }
        RCF_LAUNCH_CHECK();
        return 0;
    }
#define RCF_CALL(T)                                                                                                    \
    hipLaunchKernelGGL(resize_nhwc_fwd_kernel<T>, dim3(ew_blocks(px * (C / 4))), dim3(256), 0, rcf_stream(stream),     \
                       (const T *)x, x_pitch, (T *)y, y_pitch, N, Hi, Wi, Ho, Wo, C, align_corners,                    \
                       host_scale(Hi, Ho, align_corners), host_scale(Wi, Wo, align_corners), frame)
    RCF_DISPATCH1(dt, RCF_CALL);
#undef RCF_CALL
    RCF_LAUNCH_CHECK();
    return 0;
}
extern "C" int rcf_resize_bilinear_nhwc_fwd_f32(const float *x, int x_pitch, float *y, int y_pitch, int N, int Hi,
                                                int Wi, int Ho, int Wo, int C, int align_corners, void *stream) {
    return rcf_resize_bilinear_nhwc_fwd_mp(x, x_pitch, y, y_pitch, RCF_F32, N, Hi, Wi, Ho, Wo, C, align_corners, 0, stream);
}
extern "C" int rcf_resize_bilinear_nhwc_fwd_frame_f32(const float *x, int x_pitch, float *y, int y_pitch, int N, int Hi,
                                                      int Wi, int Ho, int Wo, int C, int align_corners, int frame,
                                                      void *stream) {
    if (frame <= 0) return RCF_EINVAL;
    return rcf_resize_bilinear_nhwc_fwd_mp(x, x_pitch, y, y_pitch, RCF_F32, N, Hi, Wi, Ho, Wo, C, align_corners, frame,
                                           stream);
}

extern "C" int rcf_resize_bilinear_nhwc_bwd_mp(const void *dy, int dy_pitch, void *dx, int dx_pitch, int dt, int beta,
                                               int N, int Hi, int Wi, int Ho, int Wo, int C, int align_corners,
                                               int frame, void *stream) {
    if (!dy || !dx || C % 4 || dy_pitch % 4 || dx_pitch % 4 || Hi <= 0 || Wi <= 0 || Ho <= 0 || Wo <= 0) return RCF_EINVAL;
    const bool general = frame == -1;                     // the whole tensor on the general kernel (see the forward)
    if (general) frame = 0;
    if (frame < 0 || (frame > 0 && (2 * frame >= Ho || 2 * frame >= Wo))) return RCF_EINVAL;
    // frame > 0: input pixels that can see the output frame lie within tc of the border (conservative: an input pixel's
    // taps lie within (1 + 1/scale) output pixels of its centre); only when the kernel accumulates (beta), otherwise
    // the interior must be written (zeros)
    int tc = 0;
    if (frame > 0 && beta) {
        const float smax = fmaxf((float)Hi / (float)Ho, (float)Wi / (float)Wo);
        tc = (int)ceilf(smax * (float)(frame + 2)) + 2;
        if (2 * tc >= Hi || 2 * tc >= Wi) tc = 0;
    }
    const long items = tc > 0 ? (long)N * (2L * tc * Wi + 2L * tc * (Hi - 2 * tc)) * (C / 4) : (long)N * Hi * Wi * (C / 4);
    if (!general && frame == 0 && !align_corners && Ho == 2 * Hi && Wo == 2 * Wi && Hi >= 2 && Wi >= 2 &&
        (long)N * ((Hi + 1) / 2) <= 65535 && (long)Wi * (C / 4) < (1L << 30)) {
        const int Hh = (Hi + 1) / 2, Wh = (Wi + 1) / 2;
        if (dt == RCF_BF16 && C % 8 == 0 && dy_pitch % 8 == 0 && dx_pitch % 8 == 0) {
            hipLaunchKernelGGL((resize2x_bwd_kernel<bf16_t, 8>), dim3(rcf_cdiv((long)Wh * (C / 8), 256), N * Hh), dim3(256), 0,
                               rcf_stream(stream), (const bf16_t *)dy, dy_pitch, (bf16_t *)dx, dx_pitch, beta, Hi, Wi, C, 0, 0);
        } else {
#define RCF_CALL(T)                                                                                                       \
    hipLaunchKernelGGL((resize2x_bwd_kernel<T, 4>), dim3(rcf_cdiv((long)Wh * (C / 4), 256), N * Hh), dim3(256), 0,         \
                       rcf_stream(stream), (const T *)dy, dy_pitch, (T *)dx, dx_pitch, beta, Hi, Wi, C, 0, 0)
            RCF_DISPATCH1(dt, RCF_CALL);
#undef RCF_CALL
        }
        RCF_LAUNCH_CHECK();
        return 0;
    }
    // the border frame of an exact 2x up-sampling, accumulating: the 2x kernel over the 2 x 2 input blocks that see the frame -- input
    // rows 0 .. frame / 2 (and their mirror images), i.e. (frame / 2 + 2) / 2 blocks, + 1 for odd sizes (a block that sees no frame
    // pixel adds nothing)
    {
        const int Hh = (Hi + 1) / 2, Wh = (Wi + 1) / 2, tb = (frame / 2 + 2) / 2 + 1;
        if (!general && frame > 0 && beta && !align_corners && Ho == 2 * Hi && Wo == 2 * Wi && 2 * tb < Hh && 2 * tb < Wh && N <= 65535 &&
            (long)Wh * Hh * (C / 4) < (1L << 30)) {
            const long Qb = 2L * tb * Wh + 2L * tb * (Hh - 2 * tb);
            if (dt == RCF_BF16 && C % 8 == 0 && dy_pitch % 8 == 0 && dx_pitch % 8 == 0) {
                hipLaunchKernelGGL((resize2x_bwd_kernel<bf16_t, 8>), dim3(rcf_cdiv(Qb * (C / 8), 256), N), dim3(256), 0,
                                   rcf_stream(stream), (const bf16_t *)dy, dy_pitch, (bf16_t *)dx, dx_pitch, beta, Hi, Wi, C, frame, tb);
            } else {
#define RCF_CALL(T)                                                                                                       \
    hipLaunchKernelGGL((resize2x_bwd_kernel<T, 4>), dim3(rcf_cdiv(Qb * (C / 4), 256), N), dim3(256), 0,                     \
                       rcf_stream(stream), (const T *)dy, dy_pitch, (T *)dx, dx_pitch, beta, Hi, Wi, C, frame, tb)
                RCF_DISPATCH1(dt, RCF_CALL);
#undef RCF_CALL
            }
            RCF_LAUNCH_CHECK();
            return 0;
        }
    }
    if ((long)N * Hi <= 65535 && (long)Wi * (C / 4) < (1L << 30)) {
        const float sh = host_scale(Hi, Ho, align_corners), sw = host_scale(Wi, Wo, align_corners);
        if (dt == RCF_BF16 && C % 8 == 0 && dy_pitch % 8 == 0 && dx_pitch % 8 == 0) {
            hipLaunchKernelGGL((resize_rows_bwd_kernel<bf16_t, 8>), dim3(rcf_cdiv((long)Wi * (C / 8), 256), N * Hi), dim3(256), 0,
                               rcf_stream(stream), (const bf16_t *)dy, dy_pitch, (bf16_t *)dx, dx_pitch, beta, Hi, Wi, Ho, Wo, C,
                               align_corners, sh, sw, frame);
        } else {
#define RCF_CALL(T)                                                                                                       \
    hipLaunchKernelGGL((resize_rows_bwd_kernel<T, 4>), dim3(rcf_cdiv((long)Wi * (C / 4), 256), N * Hi), dim3(256), 0,      \
                       rcf_stream(stream), (const T *)dy, dy_pitch, (T *)dx, dx_pitch, beta, Hi, Wi, Ho, Wo, C,             \
                       align_corners, sh, sw, frame)
            RCF_DISPATCH1(dt, RCF_CALL);
#undef RCF_CALL
        }
        RCF_LAUNCH_CHECK();
        return 0;
    }
#define RCF_CALL(T)                                                                                                  \
    hipLaunchKernelGGL(resize_nhwc_bwd_kernel<T>, dim3(ew_blocks(items)), dim3(256), 0, rcf_stream(stream),          \
                       (const T *)dy, dy_pitch, (T *)dx, dx_pitch, beta, N, Hi, Wi, Ho, Wo, C, align_corners,        \
                       host_scale(Hi, Ho, align_corners), host_scale(Wi, Wo, align_corners), frame, tc)
    RCF_DISPATCH1(dt, RCF_CALL);
#undef RCF_CALL
    RCF_LAUNCH_CHECK();
    return 0;
}
extern "C" int rcf_resize_bilinear_nhwc_bwd_f32(const float *dy, int dy_pitch, float *dx, int dx_pitch, int beta,
                                                int N, int Hi, int Wi, int Ho, int Wo, int C, int align_corners,
                                                void *stream) {
    return rcf_resize_bilinear_nhwc_bwd_mp(dy, dy_pitch, dx, dx_pitch, RCF_F32, beta, N, Hi, Wi, Ho, Wo, C, align_corners,
                                           0, stream);
}
extern "C" int rcf_resize_bilinear_nhwc_bwd_frame_f32(const float *dy, int dy_pitch, float *dx, int dx_pitch, int beta,
                                                      int N, int Hi, int Wi, int Ho, int Wo, int C, int align_corners,
                                                      int frame, void *stream) {
    if (frame <= 0) return RCF_EINVAL;
    return rcf_resize_bilinear_nhwc_bwd_mp(dy, dy_pitch, dx, dx_pitch, RCF_F32, beta, N, Hi, Wi, Ho, Wo, C, align_corners,
                                           frame, stream);
}

extern "C" int rcf_resize_bilinear_nchw_f32(const float *x, float *y, int planes, int Hi, int Wi, int Ho, int Wo,
                                            int align_corners, void *stream) {
    if (!x || !y || planes <= 0 || Hi <= 0 || Wi <= 0 || Ho <= 0 || Wo <= 0) return RCF_EINVAL;
    hipLaunchKernelGGL(resize_nchw_kernel, dim3(ew_blocks((long)planes * Ho * Wo)), dim3(256), 0, rcf_stream(stream), x,
                       y, planes, Hi, Wi, Ho, Wo, align_corners, host_scale(Hi, Ho, align_corners),
                       host_scale(Wi, Wo, align_corners));
    RCF_LAUNCH_CHECK();
    return 0;
}

extern "C" int rcf_nchw_to_nhwc_f32(const float *x, float *y, int N, int C, int H, int W, int Cpad, void *stream) {
    if (!x || !y || Cpad % 4 || Cpad < C) return RCF_EINVAL;
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(ew_blocks((long)N * H * W)), dim3(256), 0, rcf_stream(stream), x, y, N,
                       C, (long)H * W, Cpad);
    RCF_LAUNCH_CHECK();
    return 0;
}

extern "C" int rcf_nhwc_to_nchw_f32(const float *x, int x_pitch, float *y, int N, int C, int H, int W, void *stream) {
    if (!x || !y || x_pitch < C) return RCF_EINVAL;
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(ew_blocks((long)N * C * H * W)), dim3(256), 0, rcf_stream(stream), x,
                       x_pitch, y, N, C, (long)H * W);
    RCF_LAUNCH_CHECK();
    return 0;
}

extern "C" int rcf_copy2d_batched_mp(const void *src, long spitch, long sb0, long sb1, void *dst, long dpitch, long db0,
                                     long db1, int dt, long rows, int C, int beta, int n0, int n1, void *stream) {
    if (!src || !dst || C % 4 || spitch % 4 || dpitch % 4 || sb0 % 4 || sb1 % 4 || db0 % 4 || db1 % 4 || rows <= 0) return RCF_EINVAL;
    if (n0 <= 0 || n1 <= 0 || (long)n0 * n1 > 65535) return RCF_EINVAL;
    long blocks = (rows * (C / 4) + 1023) / 1024;
    const long cap = 4096 / ((long)n0 * n1) > 1 ? 4096 / ((long)n0 * n1) : 1;
    if (blocks > cap) blocks = cap;
#define RCF_CALL(T)                                                                                                     \
    hipLaunchKernelGGL(copy2d_batched_kernel<T>, dim3((unsigned)blocks, (unsigned)(n0 * n1)), dim3(256), 0,             \
                       rcf_stream(stream), (const T *)src, spitch, sb0, sb1, (T *)dst, dpitch, db0, db1, rows, C, beta, n1)
    RCF_DISPATCH1(dt, RCF_CALL);
#undef RCF_CALL
    RCF_LAUNCH_CHECK();
    return 0;
}
extern "C" int rcf_copy2d_batched_f32(const float *src, long spitch, long sb0, long sb1, float *dst, long dpitch, long db0,
                                      long db1, long rows, int C, int beta, int n0, int n1, void *stream) {
    return rcf_copy2d_batched_mp(src, spitch, sb0, sb1, dst, dpitch, db0, db1, RCF_F32, rows, C, beta, n0, n1, stream);
}

/* dst[r][c] (+)= src[r][c] with the two sides in their own storage types: strided copy, or the fp32 <-> bf16 cast */
extern "C" int rcf_copy2d_mp(const void *src, int sdt, long spitch, void *dst, int ddt, long dpitch, long rows, int C,
                             int beta, void *stream) {
    if (!src || !dst || C % 4 || spitch % 4 || dpitch % 4 || rows <= 0) return RCF_EINVAL;
    const dim3 grid(ew_blocks(rows * (C / 4)));
    hipStream_t st = rcf_stream(stream);
    if (sdt == RCF_F32 && ddt == RCF_F32)
        hipLaunchKernelGGL((copy2d_kernel<float, float>), grid, dim3(256), 0, st, (const float *)src, spitch, (float *)dst, dpitch, rows, C, beta);
    else if (sdt == RCF_BF16 && ddt == RCF_BF16)
        hipLaunchKernelGGL((copy2d_kernel<bf16_t, bf16_t>), grid, dim3(256), 0, st, (const bf16_t *)src, spitch, (bf16_t *)dst, dpitch, rows, C, beta);
    else if (sdt == RCF_F32 && ddt == RCF_BF16)
        hipLaunchKernelGGL((copy2d_kernel<float, bf16_t>), grid, dim3(256), 0, st, (const float *)src, spitch, (bf16_t *)dst, dpitch, rows, C, beta);
    else if (sdt == RCF_BF16 && ddt == RCF_F32)
        hipLaunchKernelGGL((copy2d_kernel<bf16_t, float>), grid, dim3(256), 0, st, (const bf16_t *)src, spitch, (float *)dst, dpitch, rows, C, beta);
    else
        return RCF_EINVAL;
    RCF_LAUNCH_CHECK();
    return 0;
}
extern "C" int rcf_copy2d_f32(const float *src, long spitch, float *dst, long dpitch, long rows, int C, int beta,
                              void *stream) {
    return rcf_copy2d_mp(src, RCF_F32, spitch, dst, RCF_F32, dpitch, rows, C, beta, stream);
}
