// fp32 products on the bf16 matrix pipe (gfx950): an fp32 value is exactly the sum of three bf16 pieces, a = h + m + l
// (8 significant bits each); a product a.b is taken as the six cross terms of weight >= 2^-16 (hh hm mh mm hl lh; the
// dropped ml lm ll are <= 2^-23 relative), each one v_mfma_f32_16x16x32_bf16 with fp32 accumulation.
// Shared by attn_bf16.hip (forward) and attn_bwd_bf16.hip (backward).
#pragma once
#include "common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));  // (HIP's uint4 struct sent the staging array to scratch)

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
// two floats -> packed bf16 pair, round-to-nearest-even (v_cvt_pk_bf16_f32)
__device__ __forceinline__ unsigned cvt_pk(float a, float b) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, bf16x2));
}
// (a, b) = h + m + l exactly per element; each result holds the pair's pieces as packed bf16 (a low, b high)
__device__ __forceinline__ void split3_pk(float a, float b, unsigned& h, unsigned& m, unsigned& l) {
    h = cvt_pk(a, b);
    const float ra = a - __uint_as_float(h << 16), rb = b - __uint_as_float(h & 0xFFFF0000u);
    m = cvt_pk(ra, rb);
    l = cvt_pk(ra - __uint_as_float(m << 16), rb - __uint_as_float(m & 0xFFFF0000u));
}
// eight floats -> the three bf16x8 planes
__device__ __forceinline__ void split3_x8(const float (&v)[8], bf16x8 (&pl)[3]) {
    u32x4 h, m, l;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        unsigned hh, mm, ll;
        split3_pk(v[2 * i], v[2 * i + 1], hh, mm, ll);
        h[i] = hh, m[i] = mm, l[i] = ll;
    }
    pl[0] = __builtin_bit_cast(bf16x8, h);
    pl[1] = __builtin_bit_cast(bf16x8, m);
    pl[2] = __builtin_bit_cast(bf16x8, l);
}
__device__ __forceinline__ f32x4 mfma_bf16(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
// c += a.b from the planes of a and b, smallest terms first
__device__ __forceinline__ f32x4 mfma_x6(const bf16x8 (&a)[3], const bf16x8 (&b)[3], f32x4 c) {
    c = mfma_bf16(a[2], b[0], c);
    c = mfma_bf16(a[0], b[2], c);
    c = mfma_bf16(a[1], b[1], c);
    c = mfma_bf16(a[1], b[0], c);
    c = mfma_bf16(a[0], b[1], c);
    return mfma_bf16(a[0], b[0], c);
}

// The same six terms into three accumulators by weight class (hh | hm mh | mm hl lh): the small terms are then never
// rounded at the big sum's ulp; the caller adds the classes once at the end, smallest first.
__device__ __forceinline__ void mfma_x6_3acc(const bf16x8 (&a)[3], const bf16x8 (&b)[3], f32x4& hi, f32x4& mid, f32x4& lo) {
    lo = mfma_bf16(a[2], b[0], lo);
    lo = mfma_bf16(a[0], b[2], lo);
    lo = mfma_bf16(a[1], b[1], lo);
    mid = mfma_bf16(a[1], b[0], mid);
    mid = mfma_bf16(a[0], b[1], mid);
    hi = mfma_bf16(a[0], b[0], hi);
}

// The same two products with the LDS reads of a later item's planes issued BETWEEN the MFMAs (ld(p) reads plane p): a
// wave issues in order, so reads bunched in front of an item's MFMAs are cycles in which it feeds the matrix pipe nothing.
template <class LD>
__device__ __forceinline__ f32x4 mfma_x6_il(const bf16x8 (&a)[3], const bf16x8 (&b)[3], f32x4 c, LD&& ld) {
    c = mfma_bf16(a[2], b[0], c);
    ld(0);
    __builtin_amdgcn_sched_barrier(0);
    c = mfma_bf16(a[0], b[2], c);
    c = mfma_bf16(a[1], b[1], c);
    ld(1);
    __builtin_amdgcn_sched_barrier(0);
    c = mfma_bf16(a[1], b[0], c);
    c = mfma_bf16(a[0], b[1], c);
    ld(2);
    __builtin_amdgcn_sched_barrier(0);
    return mfma_bf16(a[0], b[0], c);
}
template <class LD>
__device__ __forceinline__ void mfma_x6_3acc_il(const bf16x8 (&a)[3], const bf16x8 (&b)[3], f32x4& hi, f32x4& mid, f32x4& lo, LD&& ld) {
    lo = mfma_bf16(a[2], b[0], lo);
    ld(0);
    __builtin_amdgcn_sched_barrier(0);
    mid = mfma_bf16(a[1], b[0], mid);
    hi = mfma_bf16(a[0], b[0], hi);
    ld(1);
    __builtin_amdgcn_sched_barrier(0);
    lo = mfma_bf16(a[0], b[2], lo);
    mid = mfma_bf16(a[0], b[1], mid);
    ld(2);
    __builtin_amdgcn_sched_barrier(0);
    lo = mfma_bf16(a[1], b[1], lo);
}
