// Experiment: float atomic adds of many workgroups into one small [L][E] array (the dQ pattern of label_attn_bwd_kernel):
// device-scope adds into ONE array against workgroup-scope adds into one array PER XCD (index read from the hardware's
// XCC_ID register, so the result does not depend on how workgroups are dealt to XCDs), summed afterwards.
//   hipcc --offload-arch=gfx950 -O3 tools/atomic_scope.hip -o tools/_abl/atomic_scope && tools/_abl/atomic_scope
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ int xcc_id() {
    return (int)(__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 15);  // HW_REG_XCC_ID, bits 3:0
}

template <int MODE>  // 0: device scope, one array; 1: workgroup scope, array of this XCD; 2: device scope, array of this XCD
__global__ __launch_bounds__(256) void add_tiles(float* dst, int L, int E, int steps, int* seen) {
    const int x = xcc_id();
    if (threadIdx.x == 0) atomicOr(seen + x, 1);
    float* d = MODE == 0 ? dst : dst + (long long)x * L * E;
    const int ntile = L / 16, tile_floats = 16 * E;
    for (int it = 0; it < steps; ++it) {
        const int t = (int)(((unsigned)blockIdx.x * 2654435761u >> 8) + it) % ntile;
        float* p = d + (long long)t * tile_floats;
        for (int i = threadIdx.x; i < tile_floats; i += 256) {
            if (MODE == 1) __hip_atomic_fetch_add(p + i, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else __hip_atomic_fetch_add(p + i, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

__global__ void sum_copies(const float* src, float* out, int n, int copies) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    for (int c = 0; c < copies; ++c) s += src[(long long)c * n + i];
    out[i] = s;
}

template <int MODE>
static int run(const char* name, int L, int E, int wgs, int steps) {
    const int n = L * E, copies = MODE == 0 ? 1 : 16;
    float *dst, *out;
    int* seen;
    CHECK(hipMalloc(&dst, (size_t)copies * n * 4));
    CHECK(hipMalloc(&out, (size_t)n * 4));
    CHECK(hipMalloc(&seen, 16 * 4));
    CHECK(hipMemset(seen, 0, 64));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        CHECK(hipMemset(dst, 0, (size_t)copies * n * 4));
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(add_tiles<MODE>, dim3(wgs), dim3(256), 0, 0, dst, L, E, steps, seen);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    hipLaunchKernelGGL(sum_copies, dim3((n + 255) / 256), dim3(256), 0, 0, dst, out, n, copies);
    std::vector<float> h(n);
    CHECK(hipMemcpy(h.data(), out, (size_t)n * 4, hipMemcpyDeviceToHost));
    int hs[16];
    CHECK(hipMemcpy(hs, seen, 64, hipMemcpyDeviceToHost));
    double tot = 0;
    for (float v : h) tot += v;
    const double expect = (double)wgs * steps * 16 * E;
    int nx = 0;
    for (int i = 0; i < 16; ++i) nx += hs[i];
    const double gb = expect * 4 / 1e9;
    printf("%-34s %8.3f ms  %6.2f GB of adds  %7.1f GB/s  sum %s (%.0f / %.0f)  XCDs seen %d\n", name, best, gb, gb / (best * 1e-3),
           tot == expect ? "ok" : "WRONG", tot, expect, nx);
    (void)hipFree(dst), (void)hipFree(out), (void)hipFree(seen);
    return 0;
}

int main() {
    const int L = 9008, E = 100;  // 563 tiles of 16 rows
    for (int steps : {64, 563}) {
        printf("-- %d workgroups x %d steps of one 16 x %d tile\n", 2560, steps, E);
        if (run<0>("device scope, one array", L, E, 2560, steps)) return 1;
        if (run<2>("device scope, array per XCD", L, E, 2560, steps)) return 1;
        if (run<1>("workgroup scope, array per XCD", L, E, 2560, steps)) return 1;
    }
    return 0;
}
