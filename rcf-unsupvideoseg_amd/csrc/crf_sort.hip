// Device-wide sort / scan for the SORT-based lattice build of csrc/crf.hip (RCF_CRF_BUILD_SORT): rocPRIM's radix sort and
// inclusive scan behind two plain functions, in a file of their own (the rocPRIM templates are slow to compile and crf.hip is
// edited often).  What they replace in the reference is the hash-table insert of tools/torchCRF/src/permutohedral_gpu.cu:535-573.
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include "rcf_common.h"

// temporary storage for n (key, value) pairs / n scan items; a host-only query (no launch).  Without a device (the build
// container) rocPRIM cannot pick its configuration: a bound that covers every configuration's histograms and look-back state.
size_t rcf_crf_sort_tmp_bytes(size_t n) {
    size_t a = 0, b = 0;
    const hipError_t e1 = rocprim::radix_sort_pairs(nullptr, a, (const unsigned long long *)nullptr, (unsigned long long *)nullptr,
                                                    (const unsigned *)nullptr, (unsigned *)nullptr, n, 0, 64);
    const hipError_t e2 = rocprim::inclusive_scan(nullptr, b, (const int *)nullptr, (int *)nullptr, n, rocprim::plus<int>());
    if (e1 != hipSuccess || e2 != hipSuccess) {
        (void)hipGetLastError();
        return (size_t)(16u << 20) + n / 4;
    }
    const size_t m = a > b ? a : b;
    return m + (1u << 20);
}

int rcf_crf_sort_pairs_u64(void *tmp, size_t tmp_bytes, const unsigned long long *k_in, unsigned long long *k_out,
                           const unsigned *v_in, unsigned *v_out, size_t n, int end_bit, hipStream_t st) {
    size_t need = 0;
    hipError_t e = rocprim::radix_sort_pairs(nullptr, need, k_in, k_out, v_in, v_out, n, 0, (unsigned)end_bit, st);
    if (e != hipSuccess) return (int)e;
    if (need > tmp_bytes) return RCF_EWORKSPACE;
    e = rocprim::radix_sort_pairs(tmp, need, k_in, k_out, v_in, v_out, n, 0, (unsigned)end_bit, st);
    return e == hipSuccess ? 0 : (int)e;
}

int rcf_crf_inclusive_scan_i32(void *tmp, size_t tmp_bytes, const int *in, int *out, size_t n, hipStream_t st) {
    size_t need = 0;
    hipError_t e = rocprim::inclusive_scan(nullptr, need, in, out, n, rocprim::plus<int>(), st);
    if (e != hipSuccess) return (int)e;
    if (need > tmp_bytes) return RCF_EWORKSPACE;
    e = rocprim::inclusive_scan(tmp, need, in, out, n, rocprim::plus<int>(), st);
    return e == hipSuccess ? 0 : (int)e;
}
