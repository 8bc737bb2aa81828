// WGAN-GP step kernels that are not plain GEMMs (the Linear layers go through gzsl_gemm):
// interpolation, the closed-form gradient penalty pieces, the critic's single-output second layer, and the keyword
// head's soft-target cross entropy.  Reference: generative_module.py:22-68, decoder_module.py:39-59,
// train_gan.py:228-307.
//
// Closed-form penalty: for D(x,lab) = w2 . lrelu(W1 [x;lab] + b1) + b2 the input gradient is
//   g_b = W1[:, :D]^T (s_b * w2),  s_b = lrelu'(h_b) in {1, 0.2},  h_b = W1 [xhat_b; lab_b] + b1,
// so gp = lam * mean_b (||g_b|| - 1)^2 needs no autograd graph, and its parameter gradient touches only W1[:, :D]
// and w2 (s is piecewise constant): with G' = d gp / d g,  dW1[:, :D] += V^T G',  dw2 += sum_b (G' W1f^T)_b * s_b.
#include "common.h"
#include "../../include/gzsl_hip.h"

#define ST ((hipStream_t)stream)
#define GRID_STRIDE(i, n) \
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < (n); i += (long long)gridDim.x * blockDim.x)
static inline int ew_blocks(long long n, int per = 256) {
    long long b = (n + per - 1) / per;
    if (b > 8192) b = 8192;
    if (b < 1) b = 1;
    return (int)b;
}

__global__ void gp_interp_kernel(const float* __restrict__ real, const float* __restrict__ fake,
                                 const float* __restrict__ alpha, float* __restrict__ out, int Bg, int D) {
    const long long n = (long long)Bg * D;
    GRID_STRIDE(i, n) {
        const float a = alpha[i / D];
        out[i] = a * real[i] + (1.f - a) * fake[i];
    }
}

__global__ void gp_slope_v_kernel(const float* __restrict__ h, const float* __restrict__ w2, float* __restrict__ v,
                                  float* __restrict__ s, int Bg, int H) {
    const long long n = (long long)Bg * H;
    GRID_STRIDE(i, n) {
        const float sl = h[i] > 0.f ? 1.f : 0.2f;
        s[i] = sl;
        v[i] = sl * w2[i % H];
    }
}

__global__ void gp_rownorm_kernel(float* __restrict__ g, float* gp, int Bg, int D, float lam) {
    __shared__ float red[4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int row = blockIdx.x * (blockDim.x >> 6) + wv;
    float part = 0.f;  // this wave's share of the penalty: one atomic per block (same-address atomics cost ~100 ns each)
    for (; row < Bg; row += gridDim.x * (blockDim.x >> 6)) {
        float ss = 0.f;
        for (int d = lane; d < D; d += 64) {
            const float x = g[(long long)row * D + d];
            ss += x * x;
        }
        ss = wave_sum(ss);
        const float nrm = sqrtf(ss);
        const float coef = nrm > 0.f ? lam * 2.f * (nrm - 1.f) / ((float)Bg * nrm) : 0.f;
        for (int d = lane; d < D; d += 64) g[(long long)row * D + d] *= coef;
        part += lam * (nrm - 1.f) * (nrm - 1.f) / (float)Bg;
    }
    if (lane == 0) red[wv] = part;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(gp, red[0] + red[1] + red[2] + red[3]);
}

__global__ void gp_dw2_kernel(const float* __restrict__ dv, const float* __restrict__ s, float* dw2, int Bg, int H,
                              int rows_per_block) {
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= H) return;
    const int r0 = blockIdx.y * rows_per_block;
    const int r1 = min(Bg, r0 + rows_per_block);
    float acc = 0.f;
    for (int r = r0; r < r1; ++r) acc += dv[(long long)r * H + col] * s[(long long)r * H + col];
    atomicAdd(dw2 + col, acc);
}

// lrelu_in: `a` holds PRE-activations; LeakyReLU(0.2) is applied on the fly (the activated tensor is never stored)
__global__ void rowdot_kernel(const float* __restrict__ a, const float* __restrict__ w, const float* __restrict__ bias,
                              float* __restrict__ out, int Bg, int H, int lrelu_in) {
    const int lane = threadIdx.x & 63;
    int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    for (; row < Bg; row += gridDim.x * (blockDim.x >> 6)) {
        float s = 0.f;
        for (int j = lane; j < H; j += 64) {
            float x = a[(long long)row * H + j];
            if (lrelu_in) x = x > 0.f ? x : 0.2f * x;
            s += x * w[j];
        }
        s = wave_sum(s);
        if (lane == 0) out[row] = s + (bias ? bias[0] : 0.f);
    }
}

__global__ void weighted_sum_kernel(const float* __restrict__ x, const float* __restrict__ w, float scale, float* out,
                                    int n) {
    float part = 0.f;
    GRID_STRIDE(i, n) part += x[i] * (w ? w[i] : 1.f);
    part = wave_sum(part);
    if ((threadIdx.x & 63) == 0) atomicAdd(out, part * scale);
}

__global__ void rowdot_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ w,
                                  const float* __restrict__ act, float* __restrict__ da, float* dw, float* db, int Bg,
                                  int H, int rows_per_block, int lrelu_in) {
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    const int r0 = blockIdx.y * rows_per_block;
    const int r1 = min(Bg, r0 + rows_per_block);
    if (col < H) {
        const float wj = w[col];
        float acc = 0.f;
        for (int r = r0; r < r1; ++r) {
            float a = act[(long long)r * H + col];
            const float d = dout[r];
            acc += d * ((lrelu_in && a <= 0.f) ? 0.2f * a : a);
            if (da) da[(long long)r * H + col] = d * wj * (a > 0.f ? 1.f : 0.2f);
        }
        if (dw) atomicAdd(dw + col, acc);
    }
    if (db && blockIdx.x == 0 && threadIdx.x == 0) {
        float s = 0.f;
        for (int r = r0; r < r1; ++r) s += dout[r];
        atomicAdd(db, s);
    }
}

// One workgroup per row: log-softmax over Vk in three sweeps (max, sum, write) with the row kept in L1/L2.
__global__ void keyword_ce_kernel(float* __restrict__ logits, const int64_t* __restrict__ kw_idx,
                                  const float* __restrict__ kw_mask, const float* __restrict__ w,
                                  float* __restrict__ loss_b, int Bg, int Vk, int top_k) {
    __shared__ float red[4];
    __shared__ float bcast;
    const int row = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    float* lg = logits + (long long)row * Vk;
    float mx = -INFINITY;
    for (int v = tid; v < Vk; v += blockDim.x) mx = fmaxf(mx, lg[v]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if (lane == 0) red[wv] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float se = 0.f;
    for (int v = tid; v < Vk; v += blockDim.x) se += expf(lg[v] - mx);
    se = wave_sum(se);
    if (lane == 0) red[wv] = se;
    __syncthreads();
    se = red[0] + red[1] + red[2] + red[3];
    const float lse = mx + logf(se);
    // targets: normalised mask, assignment semantics -> slot j is live iff no later slot repeats its index
    float msum = 0.f;
    for (int j = 0; j < top_k; ++j) msum += kw_mask[(long long)row * top_k + j];
    const float inv = 1.f / (msum + 1e-7f);
    float tsum = 0.f, lsum = 0.f;
    if (tid == 0) {
        for (int j = 0; j < top_k; ++j) {
            const int64_t id = kw_idx[(long long)row * top_k + j];
            bool live = true;
            for (int j2 = j + 1; j2 < top_k; ++j2) live = live && (kw_idx[(long long)row * top_k + j2] != id);
            if (live) {
                const float t = kw_mask[(long long)row * top_k + j] * inv;
                tsum += t;
                lsum -= (lg[id] - lse) * t;
            }
        }
        loss_b[row] = lsum;
        bcast = tsum;
    }
    __syncthreads();
    tsum = bcast;
    const float wb = w ? w[row] : 1.f;
    for (int v = tid; v < Vk; v += blockDim.x) lg[v] = wb * expf(lg[v] - lse) * tsum;
    __syncthreads();
    if (tid == 0) {
        for (int j = 0; j < top_k; ++j) {
            const int64_t id = kw_idx[(long long)row * top_k + j];
            bool live = true;
            for (int j2 = j + 1; j2 < top_k; ++j2) live = live && (kw_idx[(long long)row * top_k + j2] != id);
            if (live) lg[id] -= wb * kw_mask[(long long)row * top_k + j] * inv;
        }
    }
}

extern "C" int gzsl_gp_interp(const float* real, const float* fake, const float* alpha, float* out, int Bg, int D,
                              void* stream) {
    if (Bg <= 0) return 0;
    hipLaunchKernelGGL(gp_interp_kernel, dim3(ew_blocks((long long)Bg * D)), dim3(256), 0, ST, real, fake, alpha, out, Bg,
                       D);
    GZSL_CHECK_LAUNCH();
    return 0;
}

extern "C" int gzsl_gp_slope_v(const float* h, const float* w2, float* v, float* s, int Bg, int H, void* stream) {
    if (Bg <= 0) return 0;
    hipLaunchKernelGGL(gp_slope_v_kernel, dim3(ew_blocks((long long)Bg * H)), dim3(256), 0, ST, h, w2, v, s, Bg, H);
    GZSL_CHECK_LAUNCH();
    return 0;
}

extern "C" int gzsl_gp_rownorm(float* g, float* gp, int Bg, int D, float lam, void* stream) {
    if (Bg <= 0) return 0;
    hipLaunchKernelGGL(gp_rownorm_kernel, dim3(min(ew_blocks(Bg, 4), 1024)), dim3(256), 0, ST, g, gp, Bg, D, lam);
    GZSL_CHECK_LAUNCH();
    return 0;
}

extern "C" int gzsl_gp_dw2(const float* dv, const float* s, float* dw2, int Bg, int H, void* stream) {
    if (Bg <= 0) return 0;
    const int rpb = 64;
    hipLaunchKernelGGL(gp_dw2_kernel, dim3(cdiv(H, 256), cdiv(Bg, rpb)), dim3(256), 0, ST, dv, s, dw2, Bg, H, rpb);
    GZSL_CHECK_LAUNCH();
    return 0;
}

extern "C" int gzsl_rowdot(const float* a, const float* w, const float* bias, float* out, int Bg, int H,
                           int lrelu_in, void* stream) {
    if (Bg <= 0) return 0;
    hipLaunchKernelGGL(rowdot_kernel, dim3(ew_blocks(Bg, 4)), dim3(256), 0, ST, a, w, bias, out, Bg, H, lrelu_in);
    GZSL_CHECK_LAUNCH();
    return 0;
}

extern "C" int gzsl_weighted_sum(const float* x, const float* w, float scale, float* out, int n, void* stream) {
    if (n <= 0) return 0;
    int blocks = ew_blocks(n);
    if (blocks > 256) blocks = 256;
    hipLaunchKernelGGL(weighted_sum_kernel, dim3(blocks), dim3(256), 0, ST, x, w, scale, out, n);
    GZSL_CHECK_LAUNCH();
    return 0;
}

extern "C" int gzsl_rowdot_bwd(const float* dout, const float* w, const float* act, float* da, float* dw, float* db,
                               int Bg, int H, int lrelu_in, void* stream) {
    if (Bg <= 0) return 0;
    const int rpb = 64;
    hipLaunchKernelGGL(rowdot_bwd_kernel, dim3(cdiv(H, 256), cdiv(Bg, rpb)), dim3(256), 0, ST, dout, w, act, da, dw, db,
                       Bg, H, rpb, lrelu_in);
    GZSL_CHECK_LAUNCH();
    return 0;
}

extern "C" int gzsl_keyword_ce(float* logits, const int64_t* kw_idx, const float* kw_mask, const float* w,
                               float* loss_b, int Bg, int Vk, int top_k, void* stream) {
    if (Bg <= 0) return 0;
    if (Vk <= 0 || top_k <= 0) return GZSL_EINVAL;
    hipLaunchKernelGGL(keyword_ce_kernel, dim3(Bg), dim3(256), 0, ST, logits, kw_idx, kw_mask, w, loss_b, Bg, Vk, top_k);
    GZSL_CHECK_LAUNCH();
    return 0;
}
