// Experiment: do v_mfma_f32_16x16x32_bf16 and ds_read_b128 overlap on a SIMD?  One workgroup of 8 waves per CU (two per
// SIMD), each wave loops over items of {R ds_read_b128 issued two items ahead, M MFMAs on one accumulator chain}.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_lds_overlap.hip -o tools/_abl/mfma_lds_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int R, int M>
__global__ __launch_bounds__(512) void loop_kernel(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 64 * 1024 / 4; i += 512) reinterpret_cast<float*>(lds)[i] = 1.0f / (float)(i + 1);
    __syncthreads();
    const unsigned char* base = lds + lane * 16;  // conflict-free: 64 lanes x 16 B contiguous
    bf16x8 fr[3][3];
    bf16x8 b[3];
    for (int p = 0; p < 3; ++p) b[p] = *reinterpret_cast<const bf16x8*>(base + 1024 * p);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    auto ld = [&](int slot, int it) {
#pragma unroll
        for (int p = 0; p < 3; ++p)
            if (p < R) fr[slot][p] = *reinterpret_cast<const bf16x8*>(base + ((it * 3 + p) & 31) * 1024);
    };
    if (R) { ld(0, 0); ld(1, 1); }
    else { for (int s = 0; s < 3; ++s) for (int p = 0; p < 3; ++p) fr[s][p] = b[p]; }
    for (int it = 0; it < iters; it += 3) {
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            if (R) ld((u + 2) % 3, it + u + 2);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < M; ++m) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[u][m % 3], b[(m / 2) % 3], acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (M == 0) asm volatile("" :: "v"(fr[u][0]), "v"(fr[u][1]), "v"(fr[u][2]));
        }
    }
    out[blockIdx.x * 512 + tid] = acc[0] + acc[1] + acc[2] + acc[3];
}

template <int R, int M>
static int run(const char* name, float* out) {
    const int iters = 30000, grid = 256;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((loop_kernel<R, M>), dim3(grid), dim3(512), 64 * 1024, 0, out, iters);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    // per item and SIMD (two waves): ns
    printf("%-40s %8.3f ms   %7.1f ns per item-pair on a SIMD (2 waves)\n", name, best, best * 1e6 / iters);
    return 0;
}

int main() {
    float* out;
    CHECK(hipMalloc(&out, 256 * 512 * 4));
    if (run<0, 6>("6 MFMAs, no reads", out)) return 1;
    if (run<3, 0>("3 ds_read_b128, no MFMAs", out)) return 1;
    if (run<3, 6>("3 ds_read_b128 + 6 MFMAs", out)) return 1;
    if (run<1, 6>("1 ds_read_b128 + 6 MFMAs", out)) return 1;
    if (run<3, 12>("3 ds_read_b128 + 12 MFMAs", out)) return 1;
    if (run<0, 12>("12 MFMAs, no reads", out)) return 1;
    return 0;
}
