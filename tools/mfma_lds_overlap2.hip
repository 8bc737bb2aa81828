// Experiment (follow-up of mfma_lds_overlap.hip): grow the bare {reads, MFMAs} loop toward the tile loop of
// attn_bf16.hip one ingredient at a time and watch where the overlap goes.
//   FLAGS bit 0: barrier per tile (21 items)   bit 1: ~100 VALU instructions (exp, convert) between items 7 and 8
//         bit 2: 8 global loads + 8 ds_write_b128 per thread and tile at the top   bit 3: 13 accumulators for items 8..20
//         bit 4: B operand changes per item (4 register sets)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int FLAGS>
__global__ __launch_bounds__(512) void tile_kernel(float* out, const u32x4* src, int tiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 128 * 1024 / 4; i += 512) reinterpret_cast<float*>(lds)[i] = 1.0f / (float)(i + 1);
    __syncthreads();
    bf16x8 b[4][3];
    for (int q = 0; q < 4; ++q)
        for (int p = 0; p < 3; ++p) b[q][p] = *reinterpret_cast<const bf16x8*>(lds + lane * 16 + 1024 * (3 * q + p));
    f32x4 acc[13];
    for (int j = 0; j < 13; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    u32x4 sreg[8];
    for (int k = 0; k < 8; ++k) sreg[k] = src[tid + 512 * k];
    float sv[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    float lsum = 0.f;
    for (int t = 0; t < tiles; ++t) {
        const unsigned char* base = lds + (t & 1) * 64 * 1024 + lane * 16;
        if (FLAGS & 4) {
            u32x4* dst = reinterpret_cast<u32x4*>(lds + ((t + 1) & 1) * 64 * 1024);
#pragma unroll
            for (int k = 0; k < 8; ++k) dst[tid + 512 * k] = sreg[k];
#pragma unroll
            for (int k = 0; k < 8; ++k) sreg[k] = src[(size_t)((t + 2) & 63) * 4096 + tid + 512 * k];
        }
        bf16x8 fr[2][3];
        auto ld = [&](int slot, int it) {
#pragma unroll
            for (int p = 0; p < 3; ++p) fr[slot][p] = *reinterpret_cast<const bf16x8*>(base + ((it * 3 + p) & 63) * 1024);
        };
        ld(0, 0);
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        bf16x8 pb[3] = {b[0][0], b[0][1], b[0][2]};
#pragma unroll
        for (int it = 0; it < 21; ++it) {
            if (it + 1 < 21) ld((it + 1) & 1, it + 1);
            __builtin_amdgcn_sched_barrier(0);
            const bf16x8 (&bb)[3] = it < 8 ? b[(FLAGS & 16) ? (it & 3) : 0] : pb;
            f32x4& c = it < 8 ? s : acc[(FLAGS & 8) ? it - 8 : 0];
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[it & 1][2], bb[0], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[it & 1][0], bb[2], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[it & 1][1], bb[1], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[it & 1][1], bb[0], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[it & 1][0], bb[1], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[it & 1][0], bb[0], c, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (it == 3 || it == 7) {
                for (int r = 0; r < 4; ++r) sv[4 * (it / 4) + r] = s[r];
                s = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            if (it == 7 && (FLAGS & 2)) {  // softmax-like block: 8 exp, sums, 3-way bf16 split of 8 values
                u32x4 h, m, l;
                for (int i = 0; i < 4; ++i) {
                    float p0 = __expf(sv[2 * i] * 1e-3f - 1.f), p1 = __expf(sv[2 * i + 1] * 1e-3f - 1.f);
                    lsum += p0 + p1;
                    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
                    typedef float f2 __attribute__((ext_vector_type(2)));
                    unsigned hh = __builtin_bit_cast(unsigned, __builtin_convertvector(f2{p0, p1}, bf2));
                    float r0 = p0 - __uint_as_float(hh << 16), r1 = p1 - __uint_as_float(hh & 0xFFFF0000u);
                    unsigned mm = __builtin_bit_cast(unsigned, __builtin_convertvector(f2{r0, r1}, bf2));
                    unsigned ll = __builtin_bit_cast(unsigned, __builtin_convertvector(f2{r0 - __uint_as_float(mm << 16), r1 - __uint_as_float(mm & 0xFFFF0000u)}, bf2));
                    h[i] = hh, m[i] = mm, l[i] = ll;
                }
                pb[0] = __builtin_bit_cast(bf16x8, h), pb[1] = __builtin_bit_cast(bf16x8, m), pb[2] = __builtin_bit_cast(bf16x8, l);
            }
        }
        if (FLAGS & 1) __syncthreads();
    }
    float r = lsum;
    for (int j = 0; j < 13; ++j) r += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    out[blockIdx.x * 512 + tid] = r;
}

template <int FLAGS>
static int run(const char* name, float* out, const u32x4* src) {
    const int tiles = 2000, grid = 256;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((tile_kernel<FLAGS>), dim3(grid), dim3(512), 128 * 1024, 0, out, src, tiles);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    printf("%-58s %8.3f ms   %7.1f ns per tile (21 items; 2016 MFMA cycles per wave)\n", name, best, best * 1e6 / tiles);
    return 0;
}

int main() {
    float* out;
    u32x4* src;
    CHECK(hipMalloc(&out, 256 * 512 * 4));
    CHECK(hipMalloc(&src, (size_t)66 * 4096 * 16));
    CHECK(hipMemset(src, 0, (size_t)66 * 4096 * 16));
    if (run<0>("bare tile loop", out, src)) return 1;
    if (run<1>("+ barrier per tile", out, src)) return 1;
    if (run<2>("+ softmax block", out, src)) return 1;
    if (run<4>("+ staging (8 loads + 8 ds_write_b128)", out, src)) return 1;
    if (run<8>("+ 13 accumulators", out, src)) return 1;
    if (run<16>("+ B operand per item", out, src)) return 1;
    if (run<3>("+ barrier + softmax", out, src)) return 1;
    if (run<7>("+ barrier + softmax + staging", out, src)) return 1;
    if (run<31>("all", out, src)) return 1;
    return 0;
}
