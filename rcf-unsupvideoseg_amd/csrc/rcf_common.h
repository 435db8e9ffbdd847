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
typedef __bf16 bf16_t;
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));

template <typename T> __device__ __forceinline__ f32x4 ld4(const T *p);
template <> __device__ __forceinline__ f32x4 ld4<float>(const float *p) { return *reinterpret_cast<const f32x4 *>(p); }
template <> __device__ __forceinline__ f32x4 ld4<bf16_t>(const bf16_t *p) {
    const u32x2_t r = *reinterpret_cast<const u32x2_t *>(p);
    return f32x4{__uint_as_float(r[0] << 16), __uint_as_float(r[0] & 0xffff0000u), __uint_as_float(r[1] << 16),
                 __uint_as_float(r[1] & 0xffff0000u)};
}
template <typename T> __device__ __forceinline__ void st4(T *p, f32x4 v);
template <> __device__ __forceinline__ void st4<float>(float *p, f32x4 v) { *reinterpret_cast<f32x4 *>(p) = v; }
template <> __device__ __forceinline__ void st4<bf16_t>(bf16_t *p, f32x4 v) {
    *reinterpret_cast<bf16x4_t *>(p) = __builtin_convertvector(v, bf16x4_t);
}
template <typename T> __device__ __forceinline__ float ld1(const T *p) { return (float)*p; }
template <typename T> __device__ __forceinline__ void st1(T *p, float v) { *p = (T)v; }

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
