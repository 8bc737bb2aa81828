// Experiment (not part of the library): fp32 GEMM C = A . B^T emulated on the bf16 matrix pipe by splitting every fp32
// operand into three bf16 pieces a = h + m + l (8 significant bits each, exact) and summing the cross products with fp32
// accumulation -- 9 products (all pairs) or 6 (those of weight >= 2^-16: hh hm mh mm hl lh).  Measures (1) the error of
// each variant against an fp64 reference next to a plain fp32 FMA dot product, (2) the rate of an LDS-fed tile loop on
// v_mfma_f32_16x16x32_bf16 (16 cycles per MFMA: 16x the k per cycle of v_mfma_f32_16x16x4_f32).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/bf16_split_gemm.hip -o tools/_abl/bf16_split_gemm && ./tools/_abl/bf16_split_gemm
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ uint16_t bf16_rne(float x) {
    uint32_t u = __float_as_uint(x);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
__device__ __forceinline__ float bf16_f(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }
__device__ __forceinline__ void split3(float a, uint16_t& h, uint16_t& m, uint16_t& l) {
    h = bf16_rne(a);
    const float r1 = a - bf16_f(h);
    m = bf16_rne(r1);
    const float r2 = r1 - bf16_f(m);
    l = bf16_rne(r2);
}

// fp32 [rows][K] -> three bf16 planes [3][rows][K]
__global__ void split_kernel(const float* __restrict__ x, uint16_t* __restrict__ p, long long n) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        uint16_t h, m, l;
        split3(x[i], h, m, l);
        p[i] = h, p[n + i] = m, p[2 * n + i] = l;
    }
}

__global__ void ref_fp32_kernel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int M, int N, int K) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M * N) return;
    const int m = i / N, n = i % N;
    float s = 0.f;
    for (int k = 0; k < K; ++k) s = fmaf(A[(long long)m * K + k], B[(long long)n * K + k], s);
    C[i] = s;
}

constexpr int BM = 64, BK = 32, NT = 13, BN = 16 * NT, RS = BK + 8;  // RS: bf16 elements per LDS row (80 bytes)

// A fp32 [M][K] (split on the fly at staging time), Bp = pre-split weight planes [3][N][K] bf16, C [M][N].
template <int NPROD>
__global__ __launch_bounds__(256, 2) void gemm_split_kernel(const float* __restrict__ A, const uint16_t* __restrict__ Bp,
                                                            float* __restrict__ C, int M, int N, int K) {
    __shared__ __attribute__((aligned(16))) uint16_t sA[3][BM * RS];
    __shared__ __attribute__((aligned(16))) uint16_t sB[3][BN * RS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 15, lq = lane >> 4;
    const int m0 = blockIdx.x * BM;
    f32x4 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // staging slots: A tile = 64 x 32 floats = 512 float4 (2 per thread); B tile = 3 planes x 208 rows x 4 chunks of 16 B
    constexpr int BCH = 3 * BN * 4, BPT = (BCH + 255) / 256;
    f32x4 ra[2];
    uint4 rb[BPT];
    const long long planeB = (long long)N * K;
    auto fetch = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int s = tid + 256 * i, r = s >> 3, c = (s & 7) * 4;
            ra[i] = *reinterpret_cast<const f32x4*>(A + (long long)(m0 + r) * K + k0 + c);
        }
#pragma unroll
        for (int i = 0; i < BPT; ++i) {
            const int s = min(tid + 256 * i, BCH - 1), p = s / (BN * 4), rr = (s % (BN * 4)) >> 2, c = (s & 3) * 8;
            rb[i] = *reinterpret_cast<const uint4*>(Bp + p * planeB + (long long)rr * K + k0 + c);
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int s = tid + 256 * i, r = s >> 3, c = (s & 7) * 4;
            uint16_t h[4], m[4], l[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) split3(ra[i][e], h[e], m[e], l[e]);
            *reinterpret_cast<uint2*>(&sA[0][r * RS + c]) = make_uint2(h[0] | (h[1] << 16), h[2] | (h[3] << 16));
            *reinterpret_cast<uint2*>(&sA[1][r * RS + c]) = make_uint2(m[0] | (m[1] << 16), m[2] | (m[3] << 16));
            *reinterpret_cast<uint2*>(&sA[2][r * RS + c]) = make_uint2(l[0] | (l[1] << 16), l[2] | (l[3] << 16));
        }
#pragma unroll
        for (int i = 0; i < BPT; ++i) {
            const int s = tid + 256 * i;
            if (s < BCH) {
                const int p = s / (BN * 4), rr = (s % (BN * 4)) >> 2, c = (s & 3) * 8;
                *reinterpret_cast<uint4*>(&sB[p][rr * RS + c]) = rb[i];
            }
        }
    };
    fetch(0);
    for (int k0 = 0; k0 < K; k0 += BK) {
        __syncthreads();  // the previous tile's readers are done
        stage();
        __syncthreads();
        if (k0 + BK < K) fetch(k0 + BK);
        bf16x8 fa[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) fa[p] = *reinterpret_cast<const bf16x8*>(&sA[p][(wave * 16 + lr) * RS + lq * 8]);
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            bf16x8 fb[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) fb[p] = *reinterpret_cast<const bf16x8*>(&sB[p][(j * 16 + lr) * RS + lq * 8]);
            // smallest terms first
            if (NPROD == 9) {
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[2], fb[2], acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[2], fb[1], acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[1], fb[2], acc[j], 0, 0, 0);
            }
            if (NPROD >= 6) {
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[2], fb[0], acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[0], fb[2], acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[1], fb[1], acc[j], 0, 0, 0);
            }
            if (NPROD >= 3) {
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[1], fb[0], acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[0], fb[1], acc[j], 0, 0, 0);
            }
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[0], fb[0], acc[j], 0, 0, 0);
        }
    }
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = m0 + wave * 16 + 4 * lq + r, col = 16 * j + lr;
            if (row < M && col < N) C[(long long)row * N + col] = acc[j][r];
        }
}

int main() {
    const int M = 16384, N = BN, K = 1024;
    std::vector<float> hA((size_t)M * K), hB((size_t)N * K);
    srand(1);
    auto rnd = []() { float u = 0; for (int i = 0; i < 6; ++i) u += rand() / (float)RAND_MAX; return (u - 3.f); };  // ~N(0, 0.7)
    for (auto& v : hA) v = rnd();
    for (auto& v : hB) v = rnd() * 0.1f;
    float *A, *B, *C;
    uint16_t* Bp;
    CK(hipMalloc(&A, hA.size() * 4)); CK(hipMalloc(&B, hB.size() * 4)); CK(hipMalloc(&C, (size_t)M * N * 4));
    CK(hipMalloc(&Bp, hB.size() * 6));
    CK(hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(B, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(split_kernel, dim3(1024), dim3(256), 0, 0, B, Bp, (long long)N * K);
    CK(hipDeviceSynchronize());
    // fp64 reference on 96 rows
    const int RR = 96;
    std::vector<double> ref((size_t)RR * N);
    double scale = 0;
    for (int m = 0; m < RR; ++m)
        for (int n = 0; n < N; ++n) {
            double s = 0;
            for (int k = 0; k < K; ++k) s += (double)hA[(size_t)m * K + k] * (double)hB[(size_t)n * K + k];
            ref[(size_t)m * N + n] = s;
            scale += s * s;
        }
    scale = std::sqrt(scale / ref.size());
    std::vector<float> hC((size_t)M * N);
    auto report = [&](const char* name) {
        CK(hipMemcpy(hC.data(), C, hC.size() * 4, hipMemcpyDeviceToHost));
        double mx = 0, sq = 0;
        for (size_t i = 0; i < ref.size(); ++i) {
            const double e = std::fabs((double)hC[i] - ref[i]);
            mx = e > mx ? e : mx;
            sq += e * e;
        }
        printf("%-28s max abs err %.3e  rms err %.3e  (rms |C| %.3e -> rms err / rms |C| = %.2e)\n", name, mx,
               std::sqrt(sq / ref.size()), scale, std::sqrt(sq / ref.size()) / scale);
    };
    hipLaunchKernelGGL(ref_fp32_kernel, dim3((M * N + 255) / 256), dim3(256), 0, 0, A, B, C, M, N, K);
    CK(hipDeviceSynchronize());
    report("fp32 FMA chain (k order)");
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](auto kern, const char* name) {
        hipLaunchKernelGGL(kern, dim3(M / BM), dim3(256), 0, 0, A, Bp, C, M, N, K);
        CK(hipDeviceSynchronize());
        report(name);
        const int it = 20;
        CK(hipEventRecord(e0));
        for (int i = 0; i < it; ++i) hipLaunchKernelGGL(kern, dim3(M / BM), dim3(256), 0, 0, A, Bp, C, M, N, K);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        ms /= it;
        printf("%-28s %.1f us  = %.1f fp32-equivalent TFLOP/s (2MNK)\n", name, ms * 1e3, 2.0 * M * N * K / (ms * 1e-3) / 1e12);
    };
    run(gemm_split_kernel<1>, "bf16 x1 (h.h only)");
    run(gemm_split_kernel<3>, "bf16 x3 (hh hm mh)");
    run(gemm_split_kernel<6>, "bf16 x6");
    run(gemm_split_kernel<9>, "bf16 x9");
    return 0;
}
