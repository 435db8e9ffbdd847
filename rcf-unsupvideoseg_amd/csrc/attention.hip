// Fused multi-head attention forward for the DINO ViT (models/dino_vit.py:122-133): O = softmax(scale * Q K^T) V per
// (image, head), fp32 accuracy on the bf16 matrix cores (every operand split exactly into three bf16 parts, six partial
// products -- see igemm_conv.hip), the [T, T] scores never leave the chip.
//
// Workgroup = 4 wavefronts, each owning 32 queries; the key / value sequence is walked in tiles of 64 keys staged
// (already split) in LDS for the whole workgroup.
//   S^T = K Q^T   : v_mfma_f32_32x32x16_bf16 with A = K tile rows (keys), B = the wave's Q (queries as columns), so
//                   in the accumulator layout every lane owns ONE query (column lane&31) and 16 of the 32 keys
//                   (rows (e&3) + 8(e>>2) + 4(lane>>5)) -- the online softmax (running max / sum per query) is a
//                   per-lane loop plus one exchange with lane^32.
//   O += P V      : P = exp(S - m) is re-split in registers and is, as it stands, the A operand of the second product
//                   (row = query = lane&31; the 8 values of accumulator registers 8j..8j+7 are the 8 k-elements of the
//                   lane's half).  The contraction order over keys is therefore the accumulator's row order; V is staged
//                   transposed ([dim][key]) so that the B operand supplies the same keys in the same order.
// The running rescale exp(m_old - m_new) and the final 1/l are per query, but O's accumulator rows are queries spread
// over registers, so they travel through a 32-float LDS row per wavefront.
#include "rcf_common.h"

namespace {

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
// ---- fp16 pairs (csrc/igemm_conv.hip): x * 2^k = h + m in fp16, 3 partial products.  One range serves q, k and v
// (max |qkv|); the softmax probabilities are in [0, 1] and take the fixed scale 2^14.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int h2_exponent(unsigned amax_bits) {
    const int e = (int)((amax_bits >> 23) & 0xffu);
    if (amax_bits == 0u) return 0;
    const int k = 14 - (e - 127);
    return k > 100 ? 100 : (k < -100 ? -100 : k);
}
__device__ __forceinline__ float pow2f(int k) { return __builtin_bit_cast(float, (unsigned)(127 + k) << 23); }

// the fp16-pair split of csrc/igemm_conv.hip (split2h there): h = fp16(x s), m = fp16(x s - h), through v_fma_mix{lo,hi}_f16
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

// operand fragments as raw 16 bytes; NP = 3: bf16 triples (6 products), NP = 2: fp16 pairs (3 products)
template <int NP>
__device__ __forceinline__ void split_x8(const float (&v)[8], float s, u32x4 (&f)[NP]) {
    if constexpr (NP == 2) {
        u32x2 h0, m0, h1, m1;
        split2h(f32x4{v[0], v[1], v[2], v[3]}, s, h0, m0);
        split2h(f32x4{v[4], v[5], v[6], v[7]}, s, h1, m1);
        f[0] = u32x4{h0[0], h0[1], h1[0], h1[1]};
        f[1] = u32x4{m0[0], m0[1], m1[0], m1[1]};
    } else {
        u32x2 h0, m0, l0, h1, m1, l1;
        split3(f32x4{v[0], v[1], v[2], v[3]}, h0, m0, l0);
        split3(f32x4{v[4], v[5], v[6], v[7]}, h1, m1, l1);
        f[0] = u32x4{h0[0], h0[1], h1[0], h1[1]};
        f[1] = u32x4{m0[0], m0[1], m1[0], m1[1]};
        f[2] = u32x4{l0[0], l0[1], l1[0], l1[1]};
    }
}

template <int NP>
__device__ __forceinline__ void mma_np(const u32x4 (&a)[NP], const u32x4 (&b)[NP], f32x16 &acc) {
    if constexpr (NP == 2) {
        const f16x8 a0 = __builtin_bit_cast(f16x8, a[0]), a1 = __builtin_bit_cast(f16x8, a[1]);
        const f16x8 b0 = __builtin_bit_cast(f16x8, b[0]), b1 = __builtin_bit_cast(f16x8, b[1]);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc, 0, 0, 0);
    } else {
        bf16x8 x[3], y[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            x[q] = __builtin_bit_cast(bf16x8, a[q]);
            y[q] = __builtin_bit_cast(bf16x8, b[q]);
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[2], y[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[0], y[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[1], y[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[1], y[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[0], y[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[0], y[0], acc, 0, 0, 0);
    }
}

constexpr int HD = 64;            // head dimension
constexpr int BQ = 128;           // queries per workgroup (32 per wavefront)
constexpr int BKEY = 64;          // keys per tile
constexpr int KPLANE = 4 * BKEY * 32;          // K: [group 4][key 64][16 dims] bf16 = 32 B rows (halves swizzled)
constexpr int VPITCH = BKEY * 2 + 8;           // V^T: [dim 64][64 keys] bf16, rows padded to 136 B (conflict-free b64)
constexpr int VPLANE = HD * VPITCH;

struct AttnParams {
    const float *qkv;   // [B*T][3*dim] (q | k | v, heads side by side)
    float *out;         // [B*T][dim]
    int T, nh, ld, ldo; // tokens per image, heads, row pitches of qkv / out
    float scale;
    const unsigned *amax;   // fp16 pairs: max |qkv| (raw bits); null on the bf16-triple instance
    unsigned *amax_out;     // optional: max |out| (the range of the projection GEMM that reads it)
};

template <int NP>
__global__ void __launch_bounds__(256, 2) attention_fwd_kernel(AttnParams p) {
    __shared__ __attribute__((aligned(16))) char kS[NP * KPLANE];
    __shared__ __attribute__((aligned(16))) char vS[NP * VPLANE];
    int kx = 0;                                                 // fp16 pairs: q, k, v scaled by 2^kx, P by 2^14
    if constexpr (NP == 2) kx = h2_exponent(*p.amax);
    const float sx = pow2f(kx);
    // scores in the log2 domain: the softmax is exp2(s2 - max) with s2 = s log2(e) -- one v_exp_f32 per probability instead of
    // expf's range reduction (the kernel ran 20 VALU instructions per MFMA, a third of them inside expf); the probabilities
    // are the same numbers, so O and the running sum need nothing else
    const float s_scale = (NP == 2 ? p.scale * pow2f(-kx) * pow2f(-kx) : p.scale) * 1.4426950408889634f;
    constexpr float P_SCALE = NP == 2 ? 16384.f : 1.f;
    __shared__ float rowv[4][32];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hs = lane >> 5;
    const int b = blockIdx.y / p.nh, head = blockIdx.y - b * p.nh;
    const int dim = p.nh * HD;
    const float *base = p.qkv + (long)b * p.T * p.ld + head * HD;
    const int q0 = blockIdx.x * BQ + wave * 32;                 // first query of this wavefront

    // ---- Q as the B operand of S^T = K Q^T: lane (query l31, half hs) holds dims 16g + 8hs .. +7 of its query
    u32x4 qf[4][NP];
    {
        const int q = q0 + l31;
        const bool ok = q < p.T;
        const float *qr = base + (long)(ok ? q : 0) * p.ld;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float v[8];
            const f32x4 a = ok ? *reinterpret_cast<const f32x4 *>(qr + 16 * g + 8 * hs) : f32x4{0.f, 0.f, 0.f, 0.f};
            const f32x4 c = ok ? *reinterpret_cast<const f32x4 *>(qr + 16 * g + 8 * hs + 4) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = a[e]; v[4 + e] = c[e]; }
            split_x8<NP>(v, sx, qf[g]);
        }
    }

    f32x16 oacc[2];
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) oacc[d][e] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;                       // per query (same value on lanes l and l^32)

    // loader roles.  K: thread (key kr, dim quad kq) per 16-dim group pass.  V: thread (dim quad vq, key group vg):
    // 4 keys x 4 dims, transposed in registers.
    const int kq = tid & 3, kr = tid >> 2;                      // kr 0..63
    const int k_st = kr * 32 + ((((kq >> 1) ^ ((kr >> 3) & 1))) << 4) + (kq & 1) * 8;
    const int vg = tid & 15, vq = tid >> 4;                     // vg: keys 4vg..4vg+3, vq: dims 4vq..4vq+3
    const float *kbase = base + dim, *vbase = base + 2 * dim;

    const int ntiles = (p.T + BKEY - 1) / BKEY;
    // the next tile's K / V rows travel in registers while the current tile is multiplied
    f32x4 kreg[4], vreg[4];
    auto load_tile = [&](int t) {
        const int key0 = t * BKEY;
        const int key = key0 + kr;
        const bool ok = key < p.T;
        const float *kp = kbase + (long)(ok ? key : 0) * p.ld + kq * 4;
#pragma unroll
        for (int g = 0; g < 4; ++g) kreg[g] = ok ? *reinterpret_cast<const f32x4 *>(kp + 16 * g) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int kk = key0 + 4 * vg + i;
            vreg[i] = kk < p.T ? *reinterpret_cast<const f32x4 *>(vbase + (long)kk * p.ld + 4 * vq) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    load_tile(0);
    for (int t = 0; t < ntiles; ++t) {
        const int key0 = t * BKEY;
        // ---- stage the K and V tiles (split into bf16 planes)
        __syncthreads();                                        // the previous tile's readers are done
        {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                u32x2 h, m, l;
                char *d = kS + g * (BKEY * 32) + k_st;
                if constexpr (NP == 2) {
                    split2h(kreg[g], sx, h, m);
                } else {
                    split3(kreg[g], h, m, l);
                    *reinterpret_cast<u32x2 *>(d + 2 * KPLANE) = l;
                }
                *reinterpret_cast<u32x2 *>(d) = h;
                *reinterpret_cast<u32x2 *>(d + KPLANE) = m;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {                       // dim 4vq+e: its 4 keys, k-contiguous
                u32x2 h, m, l;
                char *d = vS + (4 * vq + e) * VPITCH + vg * 8;
                if constexpr (NP == 2) {
                    split2h(f32x4{vreg[0][e], vreg[1][e], vreg[2][e], vreg[3][e]}, sx, h, m);
                } else {
                    split3(f32x4{vreg[0][e], vreg[1][e], vreg[2][e], vreg[3][e]}, h, m, l);
                    *reinterpret_cast<u32x2 *>(d + 2 * VPLANE) = l;
                }
                *reinterpret_cast<u32x2 *>(d) = h;
                *reinterpret_cast<u32x2 *>(d + VPLANE) = m;
            }
        }
        __syncthreads();
        if (t + 1 < ntiles) load_tile(t + 1);

        // ---- S^T tiles: [32 keys x 32 queries] x 2
        f32x16 sacc[2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
            for (int e = 0; e < 16; ++e) sacc[kt][e] = 0.f;
            const int swz = (hs ^ ((l31 >> 3) & 1)) << 4;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                u32x4 kf[NP];
#pragma unroll
                for (int q = 0; q < NP; ++q)
                    kf[q] = *reinterpret_cast<const u32x4 *>(kS + q * KPLANE + g * (BKEY * 32) + (kt * 32 + l31) * 32 + swz);
                mma_np<NP>(kf, qf[g], sacc[kt]);
            }
        }
        // ---- online softmax for this lane's query: 32 of the tile's 64 keys live here, 32 on lane^32
        float pv[2][16];
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int key = key0 + kt * 32 + (e & 3) + 8 * (e >> 2) + 4 * hs;
                const float s = key < p.T ? sacc[kt][e] * s_scale : -INFINITY;
                pv[kt][e] = s;
                mx = fmaxf(mx, s);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);                   // finite: every tile holds at least one real key
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);      // 0 on the first tile (m_run = -inf)
        float rs = 0.f;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int e = 0; e < 16; ++e) { pv[kt][e] = __builtin_amdgcn_exp2f(pv[kt][e] - m_new); rs += pv[kt][e]; }
        rs += __shfl_xor(rs, 32, 64);
        l_run = l_run * alpha + rs;
        m_run = m_new;
        // ---- rescale O: alpha per query -> O's rows
        if (hs == 0) rowv[wave][l31] = alpha;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float a = rowv[wave][(e & 3) + 8 * (e >> 2) + 4 * hs];
            oacc[0][e] *= a;
            oacc[1][e] *= a;
        }
        // ---- O += P V
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float v8[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) v8[i] = pv[kt][8 * j + i];
                u32x4 pf[NP];
                split_x8<NP>(v8, P_SCALE, pf);
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    u32x4 vf[NP];
                    const char *vr = vS + (d * 32 + l31) * VPITCH + (kt * 32 + 16 * j + 4 * hs) * 2;
#pragma unroll
                    for (int q = 0; q < NP; ++q) {
                        const u32x2 lo = *reinterpret_cast<const u32x2 *>(vr + q * VPLANE);          // keys +0..3
                        const u32x2 hi = *reinterpret_cast<const u32x2 *>(vr + q * VPLANE + 16);     // keys +8..11
                        vf[q] = u32x4{lo[0], lo[1], hi[0], hi[1]};
                    }
                    mma_np<NP>(pf, vf, oacc[d]);
                }
            }
    }
    // ---- O / l, store: lane holds dims d*32 + l31 of queries (e&3) + 8(e>>2) + 4hs
    __builtin_amdgcn_wave_barrier();
    if (hs == 0) rowv[wave][l31] = (NP == 2 ? pow2f(-kx - 14) : 1.0f) / l_run;      // undo the scales of P and V
    __builtin_amdgcn_wave_barrier();
    float *ob = p.out + (long)b * p.T * p.ldo + head * HD;
    unsigned omax = 0u;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int r = (e & 3) + 8 * (e >> 2) + 4 * hs;
        const int q = q0 + r;
        if (q >= p.T) continue;
        const float inv = rowv[wave][r];
        const float o0 = oacc[0][e] * inv, o1 = oacc[1][e] * inv;
        ob[(long)q * p.ldo + l31] = o0;
        ob[(long)q * p.ldo + 32 + l31] = o1;
        omax = max(omax, max(__float_as_uint(fabsf(o0)), __float_as_uint(fabsf(o1))));
    }
    if (p.amax_out) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) omax = max(omax, (unsigned)__shfl_xor((int)omax, o));
        if (lane == 0 && omax > __hip_atomic_load(p.amax_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(p.amax_out, omax);
    }
}

}  // namespace

/* out[b*T + t][h*64 + d] = sum_j softmax_j(scale * q_t . k_j) v_j[d] for every image b and head h; qkv is the output of
 * the fused qkv linear ([B*T][3*nh*64], q | k | v).  Head dimension 64 (every DINO ViT). */
extern "C" int rcf_attention_fwd_f32(const float *qkv, int ld_qkv, float *out, int ld_out, int B, int T, int nh,
                                     int head_dim, float scale, const unsigned *amax_qkv, unsigned *amax_out,
                                     void *stream) {
    if (!qkv || !out || B <= 0 || T <= 0 || nh <= 0 || head_dim != HD) return RCF_EINVAL;
    if (ld_qkv % 4 || ld_out % 4 || ld_qkv < 3 * nh * HD || ld_out < nh * HD || !rcf_aligned16(qkv) || !rcf_aligned16(out)) return RCF_EINVAL;
    if ((long)B * nh > 65535) return RCF_EINVAL;
    AttnParams p{qkv, out, T, nh, ld_qkv, ld_out, scale, amax_qkv, amax_out};
    const dim3 grid(rcf_cdiv(T, BQ), B * nh);
    if (amax_qkv) hipLaunchKernelGGL(attention_fwd_kernel<2>, grid, dim3(256), 0, rcf_stream(stream), p);   // fp16 pairs
    else hipLaunchKernelGGL(attention_fwd_kernel<3>, grid, dim3(256), 0, rcf_stream(stream), p);
    RCF_LAUNCH_CHECK();
    return 0;
}
