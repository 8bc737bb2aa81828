// fp32 MFMA GEMM with strided / windowed / batched operand addressing (gfx950).
//
// One kernel family serves every dense contraction on the hot path that is not the fused label attention:
//   conv1d as an implicit GEMM (A rows are overlapping k*E-float windows of the [B,T,E] embedding; reference
//   icd_modules.py:131 nn.Conv1d), proj+tanh (:134-135), output_proj (:159), the GRU cell GEMMs
//   (graph_module.py:122), the generator / critic MLPs (generative_module.py:48-68), the keyword head
//   (decoder_module.py:40-43) and all of their data- and weight-gradient products.
//
// C[M,N] (row-major, ldc) = act(alpha * A(M,K) . B(K,N) + bias[n]) (+ C when accumulate)
//   A(m,k) at A + (m / a_mb) * a_mbs + (k / k_inner) * a_kbs + o,  o = (m % a_mb) * a_sm + a_shift + (k % k_inner) * a_sk,
//            read as 0 unless 0 <= o < a_span   (the window check implements the conv data-gradient's zero padding)
//   B(k,n) at B + (k / k_inner) * b_kbs + (k % k_inner) * b_sk + n * b_sn
// split_k > 1: blockIdx.z owns a K range and stores alpha * partial to C + z * slab_stride (no bias/act);
//   gzsl_splitk_reduce sums the slabs in a fixed order (bitwise reproducible, no float atomics).
//
// Tiling: 256 threads = 4 waves; each wave owns WTM x WTN tiles of v_mfma_f32_16x16x4_f32 (exact fp32, 64
// FLOP/clk/SIMD - the only fp32 matrix rate gfx950 has; there is no xf32).  BK = 32, operands staged through LDS
// with the k columns of each row PERMUTED: logical k = 4s + q sits at column q*8 + s, so the four k-steps 4g..4g+3
// of lane quarter q are 16 contiguous bytes and ONE ds_read_b128 feeds four MFMAs (a ds_read_b32 per MFMA left the
// matrix pipe waiting on LDS latency).  Global loads for tile t+1 are issued into registers before tile t's MFMAs
// and written to LDS after them.
#include "common.h"
#include "../../include/gzsl_hip.h"

#define BK 32
#define LDK 36  // row stride: multiple of 4 floats so every fragment read is one aligned ds_read_b128
#define NTHREADS 256
typedef float f32x4v __attribute__((ext_vector_type(4)));

template <int ROWS, bool KCONTIG>
struct Idx {
    // element i of this thread inside a ROWS x BK tile -> (row r, k column c); lanes run along the contiguous dim
    __device__ static __forceinline__ void rc(int tid, int i, int& r, int& c) {
        int idx = tid + NTHREADS * i;
        if (KCONTIG) {
            r = idx >> 5;
            c = idx & 31;
        } else {
            r = idx % ROWS;
            c = idx / ROWS;
        }
    }
};

template <int WTM, int WTN, int WGM, int WGN, bool A_KC, bool B_KC>
__global__ __launch_bounds__(NTHREADS) void gemm_f32_kernel(GzslGemm g) {
    constexpr int BM = 16 * WTM * WGM;
    constexpr int BN = 16 * WTN * WGN;
    constexpr int NA = BM / 8;  // elements of the A tile per thread
    constexpr int NB = BN / 8;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* As = reinterpret_cast<float*>(smem_raw);
    float* Bs = As + BM * LDK;
    long long* a_base = reinterpret_cast<long long*>(Bs + BN * LDK);  // (BM+BN)*34*4 B is a multiple of 8
    int* a_o = reinterpret_cast<int*>(a_base + BM);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WGN;
    const int wn = wave % WGN;
    const int tiles_n = (g.N + BN - 1) / BN;
    const int m0 = (blockIdx.x / tiles_n) * BM;
    const int n0 = (blockIdx.x % tiles_n) * BN;
    const int z = blockIdx.z;
    int kbeg = 0, kend = g.K;
    if (g.split_k > 1) {
        kbeg = z * g.k_per_split;
        kend = min(g.K, kbeg + g.k_per_split);
    }

    for (int r = tid; r < BM; r += NTHREADS) {
        int m = m0 + r;
        long long base = 0;
        int o = 0;
        if (m < g.M) {
            int mb = m / g.a_mb;
            int mi = m - mb * g.a_mb;
            base = (long long)mb * g.a_mbs;
            o = mi * g.a_sm + g.a_shift;
        }
        a_base[r] = base;
        a_o[r] = o;
    }
    __syncthreads();

    f32x4 acc[WTM][WTN];
#pragma unroll
    for (int i = 0; i < WTM; ++i)
#pragma unroll
        for (int j = 0; j < WTN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    float ra[NA], rb[NB];
    const bool kbatched = g.k_inner < g.K;

    auto load_tile = [&](int k0) {
        int kb0 = 0, ki0 = k0;
        if (kbatched) {
            kb0 = k0 / g.k_inner;
            ki0 = k0 - kb0 * g.k_inner;
        }
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            int r, c;
            Idx<BM, A_KC>::rc(tid, i, r, c);
            int kb = kb0, ki = ki0 + c;
            if (ki >= g.k_inner) {
                ki -= g.k_inner;
                kb += 1;
            }
            int o = a_o[r] + ki * g.a_sk;
            bool ok = (m0 + r < g.M) && (k0 + c < kend) && ((unsigned)o < g.a_span);
            float v = 0.f;
            if (ok) v = g.A[a_base[r] + (long long)kb * g.a_kbs + o];
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            int r, c;
            Idx<BN, B_KC>::rc(tid, i, r, c);
            int kb = kb0, ki = ki0 + c;
            if (ki >= g.k_inner) {
                ki -= g.k_inner;
                kb += 1;
            }
            int n = n0 + r;
            bool ok = (n < g.N) && (k0 + c < kend);
            float v = 0.f;
            if (ok) v = g.B[(long long)kb * g.b_kbs + (long long)ki * g.b_sk + (long long)n * g.b_sn];
            rb[i] = v;
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            int r, c;
            Idx<BM, A_KC>::rc(tid, i, r, c);
            As[r * LDK + (c & 3) * 8 + (c >> 2)] = ra[i];
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            int r, c;
            Idx<BN, B_KC>::rc(tid, i, r, c);
            Bs[r * LDK + (c & 3) * 8 + (c >> 2)] = rb[i];
        }
    };

    const int lr = lane & 15;
    const int lq = lane >> 4;
    const float* Aw = As + (wm * WTM * 16 + lr) * LDK + lq * 8;
    const float* Bw = Bs + (wn * WTN * 16 + lr) * LDK + lq * 8;

    if (kbeg < kend) {
        load_tile(kbeg);
        store_tile();
    }
    __syncthreads();
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
        const bool more = (k0 + BK) < kend;
        if (more) load_tile(k0 + BK);
#pragma unroll
        for (int g4 = 0; g4 < BK / 16; ++g4) {
            f32x4v a[WTM], b[WTN];
#pragma unroll
            for (int i = 0; i < WTM; ++i) a[i] = *reinterpret_cast<const f32x4v*>(Aw + i * 16 * LDK + 4 * g4);
#pragma unroll
            for (int j = 0; j < WTN; ++j) b[j] = *reinterpret_cast<const f32x4v*>(Bw + j * 16 * LDK + 4 * g4);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < WTM; ++i)
#pragma unroll
                    for (int j = 0; j < WTN; ++j) acc[i][j] = mfma16(a[i][s], b[j][s], acc[i][j]);
        }
        __syncthreads();
        if (more) store_tile();
        __syncthreads();
    }

    float* C = g.C + (g.split_k > 1 ? (long long)z * g.slab_stride : 0);
    const bool plain = g.split_k > 1;
#pragma unroll
    for (int i = 0; i < WTM; ++i) {
#pragma unroll
        for (int j = 0; j < WTN; ++j) {
            int col = n0 + (wn * WTN + j) * 16 + lr;
            if (col >= g.N) continue;
            float bv = (!plain && g.bias) ? g.bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                int row = m0 + (wm * WTM + i) * 16 + lq * 4 + r;
                if (row >= g.M) continue;
                float v = g.alpha * acc[i][j][r];
                long long off = (long long)row * g.ldc + col;
                if (!plain) {
                    v += bv;
                    if (g.accumulate) v += C[off];
                    if (g.act == GZSL_ACT_RELU) v = fmaxf(v, 0.f);
                    else if (g.act == GZSL_ACT_TANH) v = tanhf(v);
                    else if (g.act == GZSL_ACT_LRELU) v = v > 0.f ? v : 0.2f * v;
                }
                C[off] = v;
            }
        }
    }
}

// out[m*ldo + n] (=|+=) act(sum_z slab[z*stride + m*N + n] + bias[n]);  optional conv-weight unpack:
// when unpack_k > 0 the slab holds dWp^T as [(j*E + e), f] and out is the reference layout [F, E, k]
// (conv_modules.0.weight, icd_modules.py:74): out[f][e][j] = sum_z slab[(j*E+e)*F + f].
__global__ void splitk_reduce_kernel(const float* __restrict__ slab, long long stride, int nsplit, float* out, int M,
                                     int N, int ldo, const float* __restrict__ bias, int act, int accumulate,
                                     int unpack_k, int unpack_e) {
    long long total = (long long)M * N;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int zz = 0; zz < nsplit; ++zz) s += slab[zz * stride + i];
        int m = (int)(i / N), n = (int)(i % N);
        if (bias) s += bias[n];
        long long o;
        if (unpack_k > 0) {
            int j = m / unpack_e, e = m % unpack_e;  // m = j*E + e, n = f
            o = ((long long)n * unpack_e + e) * unpack_k + j;
        } else {
            o = (long long)m * ldo + n;
        }
        if (accumulate) s += out[o];
        if (act == GZSL_ACT_RELU) s = fmaxf(s, 0.f);
        else if (act == GZSL_ACT_TANH) s = tanhf(s);
        else if (act == GZSL_ACT_LRELU) s = s > 0.f ? s : 0.2f * s;
        out[o] = s;
    }
}

template <int WTM, int WTN, int WGM, int WGN>
static int launch_cfg(const GzslGemm& g, hipStream_t st) {
    constexpr int BM = 16 * WTM * WGM;
    constexpr int BN = 16 * WTN * WGN;
    size_t smem = (size_t)(BM + BN) * LDK * 4 + (size_t)BM * 12;
    dim3 grid(cdiv(g.M, BM) * cdiv(g.N, BN), 1, g.split_k > 1 ? g.split_k : 1);
    const bool akc = g.a_sk == 1, bkc = g.b_sk == 1;
#define GZ_LAUNCH(AK, BK_)                                                                                    \
    hipLaunchKernelGGL((gemm_f32_kernel<WTM, WTN, WGM, WGN, AK, BK_>), grid, dim3(NTHREADS), smem, st, g)
    if (akc && bkc) GZ_LAUNCH(true, true);
    else if (akc && !bkc) GZ_LAUNCH(true, false);
    else if (!akc && bkc) GZ_LAUNCH(false, true);
    else GZ_LAUNCH(false, false);
#undef GZ_LAUNCH
    GZSL_CHECK_LAUNCH();
    return 0;
}

extern "C" int gzsl_gemm(const GzslGemm* gp, void* stream) {
    GzslGemm g = *gp;
    hipStream_t st = (hipStream_t)stream;
    if (g.M <= 0 || g.N <= 0) return 0;
    if (g.K < 0 || g.a_mb <= 0 || g.k_inner <= 0) return GZSL_EINVAL;
    if (g.k_inner < g.K && g.k_inner < BK) return GZSL_EINVAL;  // incremental k decomposition wraps at most once
    if (g.split_k > 1) {
        if (g.k_per_split <= 0 || g.k_per_split % BK != 0) return GZSL_EINVAL;
        if ((long long)g.k_per_split * g.split_k < g.K) return GZSL_EINVAL;
    }
    // pick the tile whose padded N wastes least MFMA work (ties -> wider tile)
    const int pad128 = cdiv(g.N, 128) * 128, pad208 = cdiv(g.N, 208) * 208, pad112 = cdiv(g.N, 112) * 112;
    if (pad208 < pad128 && pad208 <= pad112) return launch_cfg<2, 13, 4, 1>(g, st);
    if (pad112 < pad128) return launch_cfg<2, 7, 4, 1>(g, st);
    return launch_cfg<4, 4, 2, 2>(g, st);
}

extern "C" int gzsl_splitk_reduce(const float* slab, long long slab_stride, int nsplit, float* out, int M, int N,
                                  int ldo, const float* bias, int act, int accumulate, int unpack_k, int unpack_e,
                                  void* stream) {
    long long total = (long long)M * N;
    if (total <= 0) return 0;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, slab, slab_stride, nsplit,
                       out, M, N, ldo, bias, act, accumulate, unpack_k, unpack_e);
    GZSL_CHECK_LAUNCH();
    return 0;
}
