// Lab: what would a SORT-based lattice build cost?  (radix sort / unique / scan instead of the CAS hash table,
// /root/reference tools/torchCRF/src/permutohedral_gpu.cu:535-573 is what either replaces.)  The sort-based build sorts the tile-local
// distinct keys of all frames of a call (64-bit key = frame | 60-bit lattice key, 32-bit value = (tile, slot)), finds run heads and
// scans them into vertex ids.  This program times exactly those three device-wide steps with rocPRIM's tuned primitives on key sets
// shaped like the CRF's (8 frames: ~71 k vertices per frame, each met by ~4.5 tiles; noise frames: 2.27 M vertices, ~1.1 tiles each)
// -- a LOWER bound for any hand-written version -- to set beside the CAS build's insert kernel (0.58 ms per 8-frame call).
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void gen(unsigned long long *k, unsigned *v, long n, long distinct_per_frame, long per_frame) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const unsigned long long f = i / per_frame;
        unsigned long long h = (unsigned long long)i * 0x9E3779B97F4A7C15ull;
        h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
        const unsigned long long id = h % (unsigned long long)distinct_per_frame;      // which vertex
        unsigned long long key = id * 0x2545F4914F6CDD1Dull;                               // its 60-bit lattice key (spread)
        key &= (1ull << 60) - 1;
        k[i] = (f << 60) | key;
        v[i] = (unsigned)i;
    }
}
__global__ void heads(const unsigned long long *k, unsigned *flag, long n) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) flag[i] = (i == 0 || k[i] != k[i - 1]) ? 1u : 0u;
}

int main(int argc, char **argv) {
    struct Case { const char *name; long per_frame, distinct; } cases[] = {{"smooth frames (71 k vertices x 4.5 tiles)", 320000, 71000},
                                                                           {"noise frames (2.27 M vertices x 1.08 tiles)", 2460000, 2270000}};
    const int F = 8;
    for (auto &c : cases) {
        const long n = c.per_frame * F;
        unsigned long long *k0, *k1; unsigned *v0, *v1, *fl, *vid;
        CK(hipMalloc(&k0, n * 8)); CK(hipMalloc(&k1, n * 8)); CK(hipMalloc(&v0, n * 4)); CK(hipMalloc(&v1, n * 4));
        CK(hipMalloc(&fl, n * 4)); CK(hipMalloc(&vid, n * 4));
        gen<<<1024, 256>>>(k0, v0, n, c.distinct, c.per_frame);
        size_t tb_sort = 0, tb_scan = 0;
        CK(rocprim::radix_sort_pairs(nullptr, tb_sort, k0, k1, v0, v1, (size_t)n, 0, 63));
        CK(rocprim::exclusive_scan(nullptr, tb_scan, fl, vid, 0u, (size_t)n, rocprim::plus<unsigned>()));
        void *tmp; CK(hipMalloc(&tmp, tb_sort > tb_scan ? tb_sort : tb_scan));
        hipEvent_t e0, e1, e2, e3; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2)); CK(hipEventCreate(&e3));
        float best[3] = {1e9f, 1e9f, 1e9f};
        for (int r = 0; r < 6; ++r) {
            CK(hipEventRecord(e0));
            CK(rocprim::radix_sort_pairs(tmp, tb_sort, k0, k1, v0, v1, (size_t)n, 0, 63));
            CK(hipEventRecord(e1));
            heads<<<2048, 256>>>(k1, fl, n);
            CK(hipEventRecord(e2));
            CK(rocprim::exclusive_scan(tmp, tb_scan, fl, vid, 0u, (size_t)n, rocprim::plus<unsigned>()));
            CK(hipEventRecord(e3));
            CK(hipEventSynchronize(e3));
            float a, b, d; CK(hipEventElapsedTime(&a, e0, e1)); CK(hipEventElapsedTime(&b, e1, e2)); CK(hipEventElapsedTime(&d, e2, e3));
            if (a < best[0]) best[0] = a; if (b < best[1]) best[1] = b; if (d < best[2]) best[2] = d;
        }
        unsigned last_vid, last_fl; CK(hipMemcpy(&last_vid, vid + n - 1, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&last_fl, fl + n - 1, 4, hipMemcpyDeviceToHost));
        printf("%s: %ld (key, value) pairs of %d frames -> %u distinct: radix sort (63 bits) %.3f ms, run heads %.3f ms, scan %.3f ms; sum %.3f ms per call = %.3f ms per frame\n",
               c.name, n, F, last_vid + last_fl, best[0], best[1], best[2], best[0] + best[1] + best[2], (best[0] + best[1] + best[2]) / F);
        CK(hipFree(k0)); CK(hipFree(k1)); CK(hipFree(v0)); CK(hipFree(v1)); CK(hipFree(fl)); CK(hipFree(vid)); CK(hipFree(tmp));
    }
    return 0;
}
