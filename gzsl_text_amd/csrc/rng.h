// Counter-based dropout RNG (device statement of oracle/rng.py; see that file for the definition).
// Replaces torch's Philox stream behind nn.Dropout (reference icd_modules.py:55,59,140,157) with a hash
// keyed on GLOBAL (row, col) coordinates, so masks do not depend on tiling or on label sharding.
#pragma once
#include <stdint.h>

#define GZSL_STREAM_EMB 1u
#define GZSL_STREAM_ATTN 2u

__host__ __device__ __forceinline__ uint32_t gz_mix32(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}

__host__ __device__ __forceinline__ uint32_t gz_row_key(uint32_t seed, uint32_t stream, uint32_t row) {
    uint32_t s = gz_mix32(seed + stream * 0x632BE5ABu);
    return gz_mix32(s ^ (row * 0x9E3779B1u));
}

__host__ __device__ __forceinline__ uint32_t gz_u32(uint32_t row_key, uint32_t col) {
    return gz_mix32(row_key ^ (col * 0x85EBCA6Bu));
}

// Kernels take the drop threshold (floor(p * 2^32), computed in double on the host) and the keep
// scale 1/(1-p) as arguments; threshold 0 disables dropout (u >= 0 always keeps).
