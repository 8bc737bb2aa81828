// Measures what v_mfma_f32_16x16x4_f32 sustains on this GPU with nothing else in the loop (registers only).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o tools/_abl/mfma_peak && tools/_abl/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = a0 + threadIdx.x, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 123.456f) out[0] = s;
}

template <int NACC>
void run(int wgs_per_cu, int iters) {
    float* out;
    hipMalloc(&out, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    int grid = 256 * wgs_per_cu;
    hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(256), 0, 0, out, iters, 1.f, 2.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(256), 0, 0, out, iters, 1.f, 2.f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)grid * 4 * iters * 4 * NACC * 2048.0;
    printf("acc=%2d wg/cu=%d iters=%d: %.3f ms  %.1f TFLOP/s\n", NACC, wgs_per_cu, iters, ms, flops / ms / 1e9);
    hipFree(out);
}

int main() {
    run<13>(1, 20000);
    run<13>(2, 20000);
    run<13>(3, 20000);
    run<4>(1, 40000);
    run<4>(2, 40000);
    run<26>(1, 10000);
    run<13>(3, 200000);
    return 0;
}
