// Shared device/host helpers for the gzsl_text hot-path kernels (gfx950 / MI355X only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

// v_mfma_f32_16x16x4_f32: D = A(16x4) * B(4x16) + C, exact fp32 (k-ordered fma chain).
//   lane l: A[i = l & 15][k = l >> 4], B[k = l >> 4][j = l & 15]
//   C/D reg r of lane l: row = (l >> 4) * 4 + r, col = l & 15
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

#define GZSL_CHECK_LAUNCH()                        \
    do {                                           \
        hipError_t e__ = hipGetLastError();        \
        if (e__ != hipSuccess) return (int)e__;    \
    } while (0)

// Returned for argument errors (shape the kernels were not instantiated for, bad sizes).
#define GZSL_EINVAL 1001
// (GZSL_ETOOBIG 1002: include/gzsl_hip.h)

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// Orders LDS traffic between the lanes of ONE wave: stores issued before it are visible to loads issued after it.  LDS
// operations of a wave retire in order, so this emits no instruction; it only forbids the compiler to move the loads
// above the stores (nothing else in the source does).
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }
