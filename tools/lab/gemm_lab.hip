// Lab bench for the bf16 1x1 convs (C[M][N] = A[M][K] * W, weights K-step major [K/32][N][32]): variants of the tile kernel of
// csrc/igemm_bf16.hip, timed alone.  Not part of the product; built by tools/lab/build.sh into gpurun_out-independent binary.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <cmath>
#include <chrono>

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned OOB = 0x80000000u;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

struct P {
    const bf16_t *A; const bf16_t *Bw; bf16_t *Y;
    int M, N, K, a_pitch, y_pitch, mtiles, ntiles, mtiles8;
    int flags;        // 1 = no stores, 2 = every tile reads the rows of tile 0 (A hot in L2), 4 = no main loop, 8 = non-temporal stores,
                      // 16 = an XCD owns ONE column-tile group of every row tile (colmap), 32 = XCDs 0-3 / 4-7 own the two halves of the
                      // column tiles, 64 = every tile reads the weights of column tile 0 (W hot in L2)
    int ntile_total, nwg;
};

__device__ __forceinline__ u32x2 pack4(const f32x4 v) {
    const bf16x2 a = __builtin_convertvector(f32x2{v[0], v[1]}, bf16x2), b = __builtin_convertvector(f32x2{v[2], v[3]}, bf16x2);
    return u32x2{__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b)};
}

template <int MR, int NR>
__device__ __forceinline__ void mma_step(const char *__restrict__ As, const char *__restrict__ Bs, int arow0, int brow0, int lane, f32x16 (&acc)[MR][NR]) {
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
            for (int nr = 0; nr < NR; ++nr) acc[mr][nr] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[nr], a[mr], acc[mr][nr], 0, 0, 0);
    }
}

__device__ __forceinline__ void tile_of(int bid, int mtiles8, int ntiles, int &tile_m, int &tile_n) {
    const int grp = bid / (8 * ntiles), rem = bid - grp * 8 * ntiles;
    tile_n = rem >> 3;
    tile_m = (rem & 7) * mtiles8 + grp;
}

// epilogue of one tile: transposed accumulators -> bf16 rows (16-byte stores after a half-wave exchange)
template <int MR, int NR>
__device__ __forceinline__ void store_tile(const P &p, f32x16 (&acc)[MR][NR], int m0, int n0, int arow0, int brow0, int lane) {
    const int l31 = lane & 31, kh = lane >> 5;
#pragma unroll
    for (int nr = 0; nr < NR; ++nr) {
        const int cb = n0 + brow0 + nr * 32;
#pragma unroll
        for (int mr = 0; mr < MR; ++mr) {
            const int row = m0 + arow0 + mr * 32 + l31;
            const bool ok = row < p.M && !(p.flags & 1);
            bf16_t *yrow = p.Y + (long)row * p.y_pitch;
#pragma unroll
            for (int g = 0; g < 4; g += 2) {
                f32x4 q0 = {acc[mr][nr][4 * g], acc[mr][nr][4 * g + 1], acc[mr][nr][4 * g + 2], acc[mr][nr][4 * g + 3]};
                f32x4 q1 = {acc[mr][nr][4 * g + 4], acc[mr][nr][4 * g + 5], acc[mr][nr][4 * g + 6], acc[mr][nr][4 * g + 7]};
                u32x2 a = pack4(q0), b = pack4(q1);
                const auto r0 = __builtin_amdgcn_permlane32_swap(a[0], b[0], false, false);
                const auto r1 = __builtin_amdgcn_permlane32_swap(a[1], b[1], false, false);
                if (ok) {
                    if (p.flags & 8) __builtin_nontemporal_store(u32x4{r0[0], r1[0], r0[1], r1[1]}, reinterpret_cast<u32x4 *>(yrow + cb + 8 * (g + kh)));
                    else *reinterpret_cast<u32x4 *>(yrow + cb + 8 * (g + kh)) = u32x4{r0[0], r1[0], r0[1], r1[1]};
                }
            }
        }
    }
}

// ---- variant 0: the shipped structure (one tile per workgroup, 3 stages, LDS-DMA for both operands, pieces among the MFMAs)
template <int MR, int NR, int WM, int WN, int NST>
__global__ void __launch_bounds__(64 * WM * WN, 2) k_base(P p) {
    constexpr int NT = 64 * WM * WN, BM = 32 * MR * WM, BN = 32 * NR * WN;
    constexpr int PA = BM * 64, PB = BN * 64, STAGE = PA + PB;
    __shared__ __attribute__((aligned(16))) char smem[NST * STAGE];
    int tile_m, tile_n;
    if (p.flags & 16) {                         // XCD x owns the column tiles [x nt8, (x + 1) nt8) (ntiles % 8 == 0) of every row tile
        const int x = blockIdx.x & 7, k = blockIdx.x >> 3, nt8 = p.ntiles >> 3;
        tile_m = k / nt8;
        tile_n = x * nt8 + (k - tile_m * nt8);
    } else if (p.flags & 32) {                  // XCDs 0-3 the first half of the column tiles, 4-7 the second; each walks a quarter of the rows
        const int x = blockIdx.x & 7, k = blockIdx.x >> 3, half = p.ntiles >> 1, mt4 = (p.mtiles + 3) >> 2;
        tile_m = (x & 3) * mt4 + k / half;
        tile_n = (x >> 2) * half + k % half;
        if (k / half >= mt4) return;
    } else {
        tile_of((int)blockIdx.x, p.mtiles8, p.ntiles, tile_m, tile_n);
    }
    if (tile_m >= p.mtiles) return;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave - wm * WN;
    constexpr int ROWS = NT / 4, A_PASS = BM / ROWS, B_PASS = BN / ROWS;
    const int arow = tid >> 2;
    const int kq = (tid & 3) ^ ((arow >> 2) & 3);
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(p.A), 0, (int)OOB, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(p.Bw), 0, (int)OOB, 0x00020000);
    unsigned abase[A_PASS], bbase[B_PASS];
#pragma unroll
    for (int i = 0; i < A_PASS; ++i) {
        const int m = ((p.flags & 2) ? 0 : m0) + arow + ROWS * i;
        abase[i] = (m0 + arow + ROWS * i) < p.M ? ((unsigned)m * (unsigned)p.a_pitch + kq * 8) * 2u : OOB;
    }
#pragma unroll
    for (int i = 0; i < B_PASS; ++i) {
        const int j = ((p.flags & 64) ? 0 : n0) + arow + ROWS * i;
        bbase[i] = (n0 + arow + ROWS * i) < p.N ? (unsigned)j * 64u + (unsigned)kq * 16u : OOB;
    }
    f32x16 acc[MR][NR];
#pragma unroll
    for (int mr = 0; mr < MR; ++mr)
#pragma unroll
        for (int nr = 0; nr < NR; ++nr)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mr][nr][e] = 0.f;
    const int arow0 = wm * 32 * MR, brow0 = wn * 32 * NR;
    const int KT = (p.flags & 4) ? 0 : p.K / 32;
    const int wrow = __builtin_amdgcn_readfirstlane(wave) * 16 * 64;
    auto issue = [&](int kt, int stage) {
        char *As = smem + stage * STAGE + wrow;
        char *Bs = As + PA;
        const unsigned kb = (unsigned)kt * (unsigned)p.N * 64u;
#pragma unroll
        for (int i = 0; i < B_PASS; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (__attribute__((address_space(3))) void *)(Bs + i * ROWS * 64), 16, (int)(bbase[i] + kb), 0, 0, 0);
#pragma unroll
        for (int i = 0; i < A_PASS; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void *)(As + i * ROWS * 64), 16, (int)(abase[i] + (unsigned)kt * 64u), 0, 0, 0);
    };
    constexpr int NLD = A_PASS + B_PASS, AHEAD = NST - 1;
    auto wait_next = [&](int kt) {
        const int later = min(KT - 1, kt + AHEAD) - (kt + 1);
        if (later >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NLD) : "memory");
        else if (later == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLD) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
#pragma unroll
    for (int i = 0; i < AHEAD; ++i)
        if (i < KT) issue(i, i);
    wait_next(-1);
    __builtin_amdgcn_s_barrier();
    int st = 0, kt = 0;
    for (; kt + AHEAD < KT; ++kt) {
        const int stn = st + AHEAD >= NST ? st + AHEAD - NST : st + AHEAD;
        const char *As = smem + st * STAGE;
        mma_step<MR, NR>(As, As + PA, arow0, brow0, lane, acc);
        issue(kt + AHEAD, stn);
        __builtin_amdgcn_sched_group_barrier(0x100, 2 * (MR + NR), 0);
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 2 * MR * NR - NLD, 0);
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
    store_tile<MR, NR>(p, acc, m0, n0, arow0, brow0, lane);
}

// ---- variant 1: persistent workgroups, the tiles of a workgroup form ONE stream of K-steps through the LDS ring: the loads of the
// next tile's first K-steps are in flight while this tile's epilogue runs (2 workgroups per CU: the other one's MFMAs cover it)
template <int MR, int NR, int WM, int WN, int NST>
__global__ void __launch_bounds__(64 * WM * WN, 2) k_persist(P p) {
    constexpr int NT = 64 * WM * WN, BM = 32 * MR * WM, BN = 32 * NR * WN;
    constexpr int PA = BM * 64, PB = BN * 64, STAGE = PA + PB;
    __shared__ __attribute__((aligned(16))) char smem[NST * STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave - wm * WN;
    constexpr int ROWS = NT / 4, A_PASS = BM / ROWS, B_PASS = BN / ROWS;
    const int arow = tid >> 2;
    const int kq = (tid & 3) ^ ((arow >> 2) & 3);
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(p.A), 0, (int)OOB, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(p.Bw), 0, (int)OOB, 0x00020000);
    const int arow0 = wm * 32 * MR, brow0 = wn * 32 * NR;
    const int KT = p.K / 32;
    const int wrow = __builtin_amdgcn_readfirstlane(wave) * 16 * 64;
    // tiles of this workgroup: the contiguous range [t0, t1) of the XCD-major tile sequence (workgroup w runs on XCD w % 8)
    const int wg = (int)blockIdx.x, nwg = (int)gridDim.x;
    const int xcd = wg & 7, wx = wg >> 3, per_xcd = nwg >> 3;                 // nwg % 8 == 0
    const int tiles_x = p.mtiles8 * p.ntiles;                                 // tiles of one XCD's band (some beyond mtiles: skipped)
    const int t0 = (int)((long)tiles_x * wx / per_xcd), t1 = (int)((long)tiles_x * (wx + 1) / per_xcd);
    auto tile_mn = [&](int t, int &tm, int &tn) {        // band-local sequence: column tiles of a row tile follow each other
        const int r = t / p.ntiles;
        tn = t - r * p.ntiles;
        tm = xcd * p.mtiles8 + r;
    };
    unsigned abase[A_PASS], bbase[B_PASS];               // of the tile whose K-steps are being ISSUED
    auto set_bases = [&](int t) {
        int tm, tn;
        tile_mn(t, tm, tn);
        const bool live = t < t1 && tm < p.mtiles;
#pragma unroll
        for (int i = 0; i < A_PASS; ++i) {
            const int m = tm * BM + arow + ROWS * i;
            abase[i] = (live && m < p.M) ? ((unsigned)((p.flags & 2) ? (arow + ROWS * i) : m) * (unsigned)p.a_pitch + kq * 8) * 2u : OOB;
        }
#pragma unroll
        for (int i = 0; i < B_PASS; ++i) {
            const int j = tn * BN + arow + ROWS * i;
            bbase[i] = (live && j < p.N) ? (unsigned)j * 64u + (unsigned)kq * 16u : OOB;
        }
    };
    auto issue = [&](int kt, int stage) {
        char *As = smem + stage * STAGE + wrow;
        char *Bs = As + PA;
        const unsigned kb = (unsigned)kt * (unsigned)p.N * 64u;
#pragma unroll
        for (int i = 0; i < B_PASS; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (__attribute__((address_space(3))) void *)(Bs + i * ROWS * 64), 16, (int)(bbase[i] + kb), 0, 0, 0);
#pragma unroll
        for (int i = 0; i < A_PASS; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void *)(As + i * ROWS * 64), 16, (int)(abase[i] + (unsigned)kt * 64u), 0, 0, 0);
    };
    constexpr int NLD = A_PASS + B_PASS, AHEAD = NST - 1;
    // flat stream: step s = (tile index, kt); issue runs AHEAD steps in front of the multiply
    int it = t0, ikt = 0;                                 // next step to issue
    set_bases(it);
    int st_issue = 0;
    auto issue_next = [&]() {
        issue(ikt, st_issue);
        st_issue = st_issue == NST - 1 ? 0 : st_issue + 1;
        if (++ikt == KT) { ikt = 0; ++it; set_bases(it); }
    };
#pragma unroll
    for (int i = 0; i < AHEAD; ++i) issue_next();
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((AHEAD - 1) * NLD) : "memory");
    __builtin_amdgcn_s_barrier();
    int st = 0;
    for (int t = t0; t < t1; ++t) {
        int tm, tn;
        tile_mn(t, tm, tn);
        f32x16 acc[MR][NR];
#pragma unroll
        for (int mr = 0; mr < MR; ++mr)
#pragma unroll
            for (int nr = 0; nr < NR; ++nr)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[mr][nr][e] = 0.f;
        for (int kt = 0; kt < KT; ++kt) {
            const char *As = smem + st * STAGE;
            mma_step<MR, NR>(As, As + PA, arow0, brow0, lane, acc);
            issue_next();                                  // past the last tile: out-of-range offsets (zeros, no traffic)
            __builtin_amdgcn_sched_group_barrier(0x100, 2 * (MR + NR), 0);
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 2 * MR * NR - NLD, 0);
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((AHEAD - 1) * NLD) : "memory");
            __builtin_amdgcn_s_barrier();
            st = st == NST - 1 ? 0 : st + 1;
        }
        if (tm < p.mtiles) store_tile<MR, NR>(p, acc, tm * BM, tn * BN, arow0, brow0, lane);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}


// ---- variant 2: one tile per workgroup, SEPARATE rings: A (HBM-cold for the long-K layers) runs NSA-1 K-steps ahead, B (L2) NSB-1
template <int MR, int NR, int WM, int WN, int NSA, int NSB>
__global__ void __launch_bounds__(64 * WM * WN, 2) k_rings(P p) {
    constexpr int NT = 64 * WM * WN, BM = 32 * MR * WM, BN = 32 * NR * WN;
    constexpr int PA = BM * 64, PB = BN * 64;
    __shared__ __attribute__((aligned(16))) char smem[NSA * PA + NSB * PB];
    char *const ringA = smem, *const ringB = smem + NSA * PA;
    int tile_m, tile_n;
    if (p.flags & 16) {                         // XCD x owns the column tiles [x nt8, (x + 1) nt8) (ntiles % 8 == 0) of every row tile
        const int x = blockIdx.x & 7, k = blockIdx.x >> 3, nt8 = p.ntiles >> 3;
        tile_m = k / nt8;
        tile_n = x * nt8 + (k - tile_m * nt8);
    } else if (p.flags & 32) {                  // XCDs 0-3 the first half of the column tiles, 4-7 the second; each walks a quarter of the rows
        const int x = blockIdx.x & 7, k = blockIdx.x >> 3, half = p.ntiles >> 1, mt4 = (p.mtiles + 3) >> 2;
        tile_m = (x & 3) * mt4 + k / half;
        tile_n = (x >> 2) * half + k % half;
        if (k / half >= mt4) return;
    } else {
        tile_of((int)blockIdx.x, p.mtiles8, p.ntiles, tile_m, tile_n);
    }
    if (tile_m >= p.mtiles) return;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave - wm * WN;
    constexpr int ROWS = NT / 4, A_PASS = BM / ROWS, B_PASS = BN / ROWS;
    const int arow = tid >> 2;
    const int kq = (tid & 3) ^ ((arow >> 2) & 3);
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(p.A), 0, (int)OOB, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(p.Bw), 0, (int)OOB, 0x00020000);
    unsigned abase[A_PASS], bbase[B_PASS];
#pragma unroll
    for (int i = 0; i < A_PASS; ++i) {
        const int m = ((p.flags & 2) ? 0 : m0) + arow + ROWS * i;
        abase[i] = (m0 + arow + ROWS * i) < p.M ? ((unsigned)m * (unsigned)p.a_pitch + kq * 8) * 2u : OOB;
    }
#pragma unroll
    for (int i = 0; i < B_PASS; ++i) {
        const int j = ((p.flags & 64) ? 0 : n0) + arow + ROWS * i;
        bbase[i] = (n0 + arow + ROWS * i) < p.N ? (unsigned)j * 64u + (unsigned)kq * 16u : OOB;
    }
    f32x16 acc[MR][NR];
#pragma unroll
    for (int mr = 0; mr < MR; ++mr)
#pragma unroll
        for (int nr = 0; nr < NR; ++nr)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mr][nr][e] = 0.f;
    const int arow0 = wm * 32 * MR, brow0 = wn * 32 * NR;
    const int KT = p.K / 32;
    const int wrow = __builtin_amdgcn_readfirstlane(wave) * 16 * 64;
    const unsigned kend = (unsigned)KT;
    auto issueA = [&](int kt, int stage) {            // past the end of K: out-of-range offsets (no traffic), keeps vmcnt counts static
        char *As = ringA + stage * PA + wrow;
#pragma unroll
        for (int i = 0; i < A_PASS; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void *)(As + i * ROWS * 64), 16,
                                                     (int)((unsigned)kt < kend ? abase[i] + (unsigned)kt * 64u : OOB), 0, 0, 0);
    };
    auto issueB = [&](int kt, int stage) {
        char *Bs = ringB + stage * PB + wrow;
        const unsigned kb = (unsigned)kt * (unsigned)p.N * 64u;
#pragma unroll
        for (int i = 0; i < B_PASS; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (__attribute__((address_space(3))) void *)(Bs + i * ROWS * 64), 16,
                                                     (int)((unsigned)kt < kend ? bbase[i] + kb : OOB), 0, 0, 0);
    };
    constexpr int AA = NSA - 1, AB = NSB - 1;          // K-steps ahead
    static_assert(AA >= AB, "A runs at least as far ahead as B");
    // prologue, in the order the steady state issues: step t issues B(t + AB) then A(t + AA)
    // queue after the prologue: A(0..AA-1) interleaved with B(0..AB-1); simplest: all A first, then all B
#pragma unroll
    for (int i = 0; i < AA; ++i) issueA(i, i);
#pragma unroll
    for (int i = 0; i < AB; ++i) issueB(i, i);
    // need A(0), B(0): everything but B(1..AB-1) -- conservative: wait for all but the last (AB-1)*B_PASS
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((AB - 1) * B_PASS) : "memory");
    __builtin_amdgcn_s_barrier();
    int sa = 0, sb = 0;
    for (int kt = 0; kt < KT; ++kt) {
        const int san = sa + AA >= NSA ? sa + AA - NSA : sa + AA, sbn = sb + AB >= NSB ? sb + AB - NSB : sb + AB;
        mma_step<MR, NR>(ringA + sa * PA, ringB + sb * PB, arow0, brow0, lane, acc);
        issueB(kt + AB, sbn);
        issueA(kt + AA, san);
        __builtin_amdgcn_sched_group_barrier(0x100, 2 * (MR + NR), 0);
#pragma unroll
        for (int i = 0; i < A_PASS + B_PASS; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 2 * MR * NR - A_PASS - B_PASS, 0);
        // next step needs A(kt+1), B(kt+1).  Queue tail (oldest first): ... B(kt+1) A(kt+AA-?)...  Steady-state order per step s: B(s+AB), A(s+AA).
        // Outstanding groups younger than B(kt+1) [issued at step kt+1-AB]: A(kt+1-AB+AA) and the full groups of steps kt+2-AB .. kt.
        // A(kt+1) was issued at step kt+1-AA <= kt+1-AB: older.  So allowed outstanding = A_PASS + (AB-1) * (A_PASS + B_PASS).
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((AA == AB ? 0 : A_PASS) + (AB - 1) * (A_PASS + B_PASS)) : "memory");
        __builtin_amdgcn_s_barrier();
        sa = sa == NSA - 1 ? 0 : sa + 1;
        sb = sb == NSB - 1 ? 0 : sb + 1;
    }
    store_tile<MR, NR>(p, acc, m0, n0, arow0, brow0, lane);
}

// ---- variant 5: TWO TEAMS of 4 waves in one 512-thread workgroup (1 per CU), each team a persistent stream of 128x256 tiles through its
// own 3-stage ring; a tile = KT multiply slots + 8 epilogue slots, every slot ends in ONE workgroup barrier, team 1 runs half a tile
// period behind team 0: while one team stores (VALU + memory pipe) the other multiplies (matrix pipe) on the same SIMDs.
template <int MR, int NR>
__device__ __forceinline__ void store_chunk(const P &p, f32x16 (&acc)[MR][NR], int m0, int n0, int arow0, int brow0, int lane, int nr, int gp) {
    const int l31 = lane & 31, kh = lane >> 5;
    const int cb = n0 + brow0 + nr * 32, g = 2 * gp;
#pragma unroll
    for (int mr = 0; mr < MR; ++mr) {
        const int row = m0 + arow0 + mr * 32 + l31;
        const bool ok = row < p.M && !(p.flags & 1);
        bf16_t *yrow = p.Y + (long)row * p.y_pitch;
        f32x4 q0 = {acc[mr][nr][4 * g], acc[mr][nr][4 * g + 1], acc[mr][nr][4 * g + 2], acc[mr][nr][4 * g + 3]};
        f32x4 q1 = {acc[mr][nr][4 * g + 4], acc[mr][nr][4 * g + 5], acc[mr][nr][4 * g + 6], acc[mr][nr][4 * g + 7]};
        u32x2 a = pack4(q0), b = pack4(q1);
        const auto r0 = __builtin_amdgcn_permlane32_swap(a[0], b[0], false, false);
        const auto r1 = __builtin_amdgcn_permlane32_swap(a[1], b[1], false, false);
        if (ok) *reinterpret_cast<u32x4 *>(yrow + cb + 8 * (g + kh)) = u32x4{r0[0], r1[0], r0[1], r1[1]};
    }
}

__global__ void __launch_bounds__(512, 1) k_teams(P p) {
    constexpr int MR = 2, NR = 4, WM = 2, WN = 2, NST = 3;
    constexpr int NT = 256, BM = 128, BN = 256;
    constexpr int PA = BM * 64, PB = BN * 64, STAGE = PA + PB, E = 8;
    __shared__ __attribute__((aligned(16))) char smem_all[2 * NST * STAGE];
    const int team = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8);
    char *smem = smem_all + team * (NST * STAGE);
    const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave - wm * WN;
    constexpr int ROWS = NT / 4, A_PASS = BM / ROWS, B_PASS = BN / ROWS;
    const int arow = tid >> 2;
    const int kq = (tid & 3) ^ ((arow >> 2) & 3);
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(p.A), 0, (int)OOB, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(p.Bw), 0, (int)OOB, 0x00020000);
    const int arow0 = wm * 32 * MR, brow0 = wn * 32 * NR;
    const int KT = p.K / 32;
    const int wrow = __builtin_amdgcn_readfirstlane(wave) * 16 * 64;
    const int wg = (int)blockIdx.x, nwg = (int)gridDim.x;
    const int xcd = wg & 7, wx = wg >> 3, per_xcd = nwg >> 3;
    const int tiles_x = p.mtiles8 * p.ntiles;
    const int t0 = (int)((long)tiles_x * wx / per_xcd), t1 = (int)((long)tiles_x * (wx + 1) / per_xcd);
    auto tile_mn = [&](int t, int &tm, int &tn) {
        const int r = t / p.ntiles;
        tn = t - r * p.ntiles;
        tm = xcd * p.mtiles8 + r;
    };
    unsigned abase[A_PASS], bbase[B_PASS];
    auto set_bases = [&](int t) {
        int tm, tn;
        tile_mn(t, tm, tn);
        const bool live = t < t1 && tm < p.mtiles;
#pragma unroll
        for (int i = 0; i < A_PASS; ++i) {
            const int m = tm * BM + arow + ROWS * i;
            abase[i] = (live && m < p.M) ? ((unsigned)((p.flags & 2) ? (arow + ROWS * i) : m) * (unsigned)p.a_pitch + kq * 8) * 2u : OOB;
        }
#pragma unroll
        for (int i = 0; i < B_PASS; ++i) {
            const int j = tn * BN + arow + ROWS * i;
            bbase[i] = (live && j < p.N) ? (unsigned)j * 64u + (unsigned)kq * 16u : OOB;
        }
    };
    auto issue = [&](int kt, int stage) {
        char *As = smem + stage * STAGE + wrow;
        char *Bs = As + PA;
        const unsigned kb = (unsigned)kt * (unsigned)p.N * 64u;
#pragma unroll
        for (int i = 0; i < B_PASS; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (__attribute__((address_space(3))) void *)(Bs + i * ROWS * 64), 16, (int)(bbase[i] + kb), 0, 0, 0);
#pragma unroll
        for (int i = 0; i < A_PASS; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void *)(As + i * ROWS * 64), 16, (int)(abase[i] + (unsigned)kt * 64u), 0, 0, 0);
    };
    constexpr int NLD = A_PASS + B_PASS, AHEAD = NST - 1;
    constexpr int STORES = 2 * MR * NR;                // 16-byte stores per lane and tile
    int it = t0 + team, ikt = 0, st_issue = 0;             // next step to issue (this team's tiles: t0 + team, + 2, ...)
    set_bases(it);
    auto issue_next = [&]() {
        issue(ikt, st_issue);
        st_issue = st_issue == NST - 1 ? 0 : st_issue + 1;
        if (++ikt == KT) { ikt = 0; it += 2; set_bases(it); }
    };
#pragma unroll
    for (int i = 0; i < AHEAD; ++i) issue_next();
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((AHEAD - 1) * NLD) : "memory");
    __builtin_amdgcn_s_barrier();
    if (team == 1) {
        const int H = (KT + E) / 2;
        for (int i = 0; i < H; ++i) __builtin_amdgcn_s_barrier();
    }
    int st = 0;
    bool prev_stores = false;
    for (int t = t0 + team; t < t1; t += 2) {
        int tm, tn;
        tile_mn(t, tm, tn);
        f32x16 acc[MR][NR];
#pragma unroll
        for (int mr = 0; mr < MR; ++mr)
#pragma unroll
            for (int nr = 0; nr < NR; ++nr)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[mr][nr][e] = 0.f;
        for (int kt = 0; kt < KT; ++kt) {
            const char *As = smem + st * STAGE;
            mma_step<MR, NR>(As, As + PA, arow0, brow0, lane, acc);
            issue_next();
            __builtin_amdgcn_sched_group_barrier(0x100, 2 * (MR + NR), 0);
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 2 * MR * NR - NLD, 0);
            // the previous tile's stores (older than everything issued in this tile) may still be in flight in the first slots:
            // in-order completion, so the counts below only ever wait for what the next slot needs
            // (only when the previous tile did issue all of its stores: not the first tile, not after a partial edge tile)
            if (kt < 2 && prev_stores) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((AHEAD - 1) * NLD + STORES) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((AHEAD - 1) * NLD) : "memory");
            __builtin_amdgcn_s_barrier();
            st = st == NST - 1 ? 0 : st + 1;
        }
        const bool live = tm < p.mtiles;
        prev_stores = live && (tm + 1) * BM <= p.M && !(p.flags & 1);
#pragma unroll
        for (int ch = 0; ch < E; ++ch) {
            if (live) store_chunk<MR, NR>(p, acc, tm * BM, tn * BN, arow0, brow0, lane, ch >> 1, ch & 1);
            __builtin_amdgcn_s_barrier();
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// the epilogue's stores alone, WITHOUT the tile kernel's 72 KB of LDS (so that its workgroups fit beside two main-loop workgroups per CU)
__global__ void __launch_bounds__(256) k_store_only(P p) {
    int tile_m, tile_n;
    tile_of((int)blockIdx.x, p.mtiles8, p.ntiles, tile_m, tile_n);
    if (tile_m >= p.mtiles) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    f32x16 acc[2][4];
#pragma unroll
    for (int mr = 0; mr < 2; ++mr)
#pragma unroll
        for (int nr = 0; nr < 4; ++nr)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mr][nr][e] = (float)(lane + e);
    store_tile<2, 4>(p, acc, tile_m * 128, tile_n * 256, wm * 64, wn * 128, lane);
}

__global__ void ref_kernel(const bf16_t *A, const bf16_t *Bw, float *out, int M, int N, int K, int a_pitch, const int *rows, int nrows) {
    const int r = blockIdx.x, row = rows[r];
    for (int c = threadIdx.x; c < N; c += blockDim.x) {
        float s = 0.f;
        for (int k = 0; k < K; ++k) s += (float)A[(long)row * a_pitch + k] * (float)Bw[((long)(k >> 5) * N + c) * 32 + (k & 31)];
        out[(long)r * N + c] = s;
    }
}
__global__ void fill_kernel(bf16_t *p, long n, unsigned seed, float scale) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u ^ seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
        // roughly normal: sum of 4 uniform bytes
        const float u = ((h & 255) + ((h >> 8) & 255) + ((h >> 16) & 255) + (h >> 24)) / 255.f - 2.f;
        p[i] = (bf16_t)(u * scale);
    }
}
__global__ void flush_kernel(float *p, long n) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = p[i] * 1.0001f + 1.f;
}

int main(int argc, char **argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 102720, K = argc > 2 ? atoi(argv[2]) : 256, N = argc > 3 ? atoi(argv[3]) : 1024;
    const int reps = argc > 4 ? atoi(argv[4]) : 20;
    bf16_t *A, *W, *Y;
    CK(hipMalloc(&A, (size_t)M * K * 2));
    CK(hipMalloc(&W, (size_t)K * N * 2));
    CK(hipMalloc(&Y, (size_t)M * N * 2));
    float *flush;
    const long nflush = 160L << 20;      // 640 MB: larger than the Infinity Cache
    CK(hipMalloc(&flush, nflush * 4));
    CK(hipMemset(flush, 0, nflush * 4));
    fill_kernel<<<1024, 256>>>(A, (long)M * K, 1u, 1.f);
    fill_kernel<<<1024, 256>>>(W, (long)K * N, 2u, 0.05f);
    CK(hipDeviceSynchronize());
    const int nchk = 64;
    std::vector<int> rows(nchk);
    for (int i = 0; i < nchk; ++i) rows[i] = (int)(((long)i * 7919 * 131) % M);
    rows[0] = 0; rows[1] = M - 1; rows[2] = 127; rows[3] = 128;
    int *drows; float *dref;
    CK(hipMalloc(&drows, nchk * 4));
    CK(hipMalloc(&dref, (size_t)nchk * N * 4));
    CK(hipMemcpy(drows, rows.data(), nchk * 4, hipMemcpyHostToDevice));
    ref_kernel<<<nchk, 256>>>(A, W, dref, M, N, K, K, drows, nchk);
    std::vector<float> ref((size_t)nchk * N);
    CK(hipMemcpy(ref.data(), dref, ref.size() * 4, hipMemcpyDeviceToHost));
    std::vector<bf16_t> yrow(N);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    constexpr int BM = 128, BN = 256;
    P p{A, W, Y, M, N, K, K, N, (M + BM - 1) / BM, (N + BN - 1) / BN, 0, 0, 0, 0};
    p.mtiles8 = (p.mtiles + 7) / 8;
    const double gf = 2.0 * M * N * K / 1e9, mb = ((double)M * K * 2 + (double)M * N * 2 + (double)K * N * 2) / 1e6;
    printf("M %d K %d N %d: %.1f GF, %.1f MB algorithmic; tiles %d x %d\n", M, K, N, gf, mb, p.mtiles, p.ntiles);
    struct V { const char *name; int kind; int flags; int nwg; };
    std::vector<V> vs = {
        {"base", 0, 0, 0}, {"base nostore", 0, 1, 0}, {"base hotA", 0, 2, 0}, {"base hotA nostore", 0, 3, 0}, {"base nomma(store only)", 0, 4, 0},
        {"base hotW", 0, 64, 0}, {"base hotA hotW", 0, 66, 0}, {"base nt-stores", 0, 8, 0}, {"base hotA nt-stores", 0, 10, 0},
        {"colmap (XCD = col tiles)", 0, 16, 0}, {"colmap nt-stores", 0, 24, 0}, {"halves (4 XCDs per col half)", 0, 32, 0}, {"halves nt-stores", 0, 40, 0},

        {"teams 256wg", 5, 0, 256}, {"teams 256wg nostore", 5, 1, 256}, {"teams hotA", 5, 2, 256},
        {"persist 512wg", 1, 0, 512},
    };
    for (auto &v : vs) {
        p.flags = v.flags;
        auto launch = [&]() {
            if (v.kind == 0 && (v.flags & 16) && p.ntiles % 8) return;
            if (v.kind == 0 && (v.flags & 32) && p.ntiles % 2) return;
            if (v.kind == 0 && (v.flags & 32)) k_base<2, 4, 2, 2, 3><<<8 * ((p.mtiles + 3) / 4) * (p.ntiles / 2), 256>>>(p);
            else if (v.kind == 0 && (v.flags & 16)) k_base<2, 4, 2, 2, 3><<<p.mtiles * p.ntiles, 256>>>(p);
            else if (v.kind == 0) k_base<2, 4, 2, 2, 3><<<8 * p.mtiles8 * p.ntiles, 256>>>(p);
            else if (v.kind == 5) k_teams<<<v.nwg, 512>>>(p);
            else if (v.kind == 2) k_rings<2, 4, 2, 2, 4, 3><<<8 * p.mtiles8 * p.ntiles, 256>>>(p);
            else if (v.kind == 3) k_rings<2, 4, 2, 2, 5, 2><<<8 * p.mtiles8 * p.ntiles, 256>>>(p);
            else if (v.kind == 4) k_rings<2, 4, 2, 2, 3, 3><<<8 * p.mtiles8 * p.ntiles, 256>>>(p);
            else k_persist<2, 4, 2, 2, 3><<<v.nwg, 256>>>(p);
        };
        CK(hipMemset(Y, 0, (size_t)M * N * 2));
        launch();
        CK(hipDeviceSynchronize());
        double maxerr = 0;
        if (!(v.flags & (7 | 64))) {
            for (int i = 0; i < nchk; ++i) {
                CK(hipMemcpy(yrow.data(), Y + (long)rows[i] * N, N * 2, hipMemcpyDeviceToHost));
                for (int c = 0; c < N; ++c) {
                    const double d = fabs((double)(float)yrow[c] - ref[(size_t)i * N + c]) / (fabs(ref[(size_t)i * N + c]) + 1.0);
                    if (d > maxerr) maxerr = d;
                }
            }
        }
        // back to back
        CK(hipEventRecord(e0));
        for (int r = 0; r < reps; ++r) launch();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double t_bb = ms / reps;
        // with the caches flushed before every launch
        double t_fl = 0;
        for (int r = 0; r < 5; ++r) {
            flush_kernel<<<2048, 256>>>(flush, nflush);
            CK(hipEventRecord(e0));
            launch();
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
            t_fl += ms / 5;
        }
        printf("%-28s  back-to-back %.4f ms (%.0f TF/s, %.2f TB/s)   flushed %.4f ms (%.0f TF/s, %.2f TB/s)   maxrelerr %.2e\n", v.name, t_bb, gf / t_bb, mb / t_bb / 1e3,
               t_fl, gf / t_fl, mb / t_fl / 1e3, maxerr);
    }
    {
        // can the two phases overlap AT ALL on this chip?  the main loop alone (nostore) and the stores alone (nomma), launched
        // together on two streams, against each of them alone and against their sum
        hipStream_t sa, sb;
        CK(hipStreamCreate(&sa)); CK(hipStreamCreate(&sb));
        P pa = p, pb = p;
        pa.flags = 1; pb.flags = 0;
        const dim3 grid(8 * p.mtiles8 * p.ntiles);
        auto wall = [&](int which) {
            CK(hipDeviceSynchronize());
            auto t0 = std::chrono::high_resolution_clock::now();
            for (int r = 0; r < reps; ++r) {
                if (which & 1) k_base<2, 4, 2, 2, 3><<<grid, 256, 0, sa>>>(pa);
                if (which & 2) k_store_only<<<grid, 256, 0, sb>>>(pb);
            }
            CK(hipDeviceSynchronize());
            return std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - t0).count() / reps;
        };
        wall(3);
        const double ta = wall(1), tb = wall(2), tab = wall(3);
        printf("two streams: main loop alone %.4f ms, stores alone %.4f ms, both at once %.4f ms (sum %.4f, max %.4f)\n", ta, tb, tab, ta + tb, ta > tb ? ta : tb);
    }
    return 0;
}
