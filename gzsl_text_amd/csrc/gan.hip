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

static inline bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }

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

// One wave per row; with V = 4 (D % 4 == 0, 16-byte aligned rows) a row of D <= 1024 floats stays in registers between
// the norm and the scaling pass (one read, one write).
template <int V>
__global__ __launch_bounds__(256) void gp_rownorm_kernel(float* __restrict__ g, float* gp, int Bg, int D, float lam) {
    __shared__ float red[4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int row = blockIdx.x * 4 + wv;
    float part = 0.f;  // this wave's share of the penalty: one atomic per block (same-address atomics cost ~100 ns each)
    for (; row < Bg; row += gridDim.x * 4) {
        float* gr = g + (long long)row * D;
        float ss = 0.f;
        f32x4 keep[4];
        if (V == 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int d = (lane + 64 * u) * 4;
                keep[u] = d < D ? *reinterpret_cast<const f32x4*>(gr + d) : f32x4{0.f, 0.f, 0.f, 0.f};
                ss += keep[u][0] * keep[u][0] + keep[u][1] * keep[u][1] + keep[u][2] * keep[u][2] + keep[u][3] * keep[u][3];
            }
            for (int d = (lane + 256) * 4; d < D; d += 256)
#pragma unroll
                for (int k = 0; k < 4; ++k) ss += gr[d + k] * gr[d + k];
        } else {
            for (int d = lane; d < D; d += 64) ss += gr[d] * gr[d];
        }
        ss = wave_sum(ss);
        const float nrm = sqrtf(ss);
        const float coef = nrm > 0.f ? lam * 2.f * (nrm - 1.f) / ((float)Bg * nrm) : 0.f;
        if (V == 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int d = (lane + 64 * u) * 4;
                if (d < D) *reinterpret_cast<f32x4*>(gr + d) = keep[u] * coef;
            }
            for (int d = (lane + 256) * 4; d < D; d += 256)
#pragma unroll
                for (int k = 0; k < 4; ++k) gr[d + k] *= coef;
        } else {
            for (int d = lane; d < D; d += 64) gr[d] *= coef;
        }
        part += lam * (nrm - 1.f) * (nrm - 1.f) / (float)Bg;
    }
    if (lane == 0) red[wv] = part;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(gp, red[0] + red[1] + red[2] + red[3]);
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

// Column-owner layout of the head kernels below: a lane owns V adjacent columns and walks the rows of its row block
// (interleaved over the block's waves); dw stays in registers until one atomic per column and row block.
template <int V>
__global__ __launch_bounds__(1024) void rowdot_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ w,
                                                          const float* __restrict__ act, float* __restrict__ da, float* dw,
                                                          float* db, int Bg, int H, int rb, int lrelu_in) {
    __shared__ float red[16][64 * V];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int col = (blockIdx.x * 64 + lane) * V;
    const bool live = col < H;
    const int r0 = blockIdx.y * rb, r1 = min(Bg, r0 + rb);
    float wj[V], acc[V];
#pragma unroll
    for (int i = 0; i < V; ++i) wj[i] = live ? w[col + i] : 0.f, acc[i] = 0.f;
    if (live) {
        for (int r = r0 + wv; r < r1; r += nw) {
            const float d = dout[r];
            const long long o = (long long)r * H + col;
            float a[V], g[V];
            if (V == 4) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(act + o);
#pragma unroll
                for (int i = 0; i < V; ++i) a[i] = t[i];
            } else {
#pragma unroll
                for (int i = 0; i < V; ++i) a[i] = act[o + i];
            }
#pragma unroll
            for (int i = 0; i < V; ++i) {
                acc[i] += d * ((lrelu_in && a[i] <= 0.f) ? 0.2f * a[i] : a[i]);
                g[i] = d * wj[i] * (a[i] > 0.f ? 1.f : 0.2f);
            }
            if (da) {
                if (V == 4) *reinterpret_cast<f32x4*>(da + o) = f32x4{g[0], g[1 % V], g[2 % V], g[3 % V]};
                else
#pragma unroll
                    for (int i = 0; i < V; ++i) da[o + i] = g[i];
            }
        }
    }
    if (dw) {
#pragma unroll
        for (int i = 0; i < V; ++i) red[wv][lane * V + i] = acc[i];
        __syncthreads();
        if (wv == 0 && live) {
#pragma unroll
            for (int i = 0; i < V; ++i) {
                float t = 0.f;
                for (int u = 0; u < nw; ++u) t += red[u][lane * V + i];
                atomicAdd(dw + col + i, t);
            }
        }
    }
    if (db && blockIdx.x == 0 && threadIdx.x == 0) {
        float s = 0.f;
        for (int r = r0; r < r1; ++r) s += dout[r];
        atomicAdd(db, s);
    }
}

// One workgroup per row: log-softmax over Vk.  REG (Vk % 4 == 0, Vk <= 8192, aligned rows): the row is read once into
// registers (8 float4 per lane) and written once; otherwise three sweeps (max, sum, write) with the row kept in L1/L2.
template <bool REG>
__global__ __launch_bounds__(256) void keyword_ce_kernel(float* __restrict__ logits, const int64_t* __restrict__ kw_idx,
                                                         const float* __restrict__ kw_mask, const float* __restrict__ w,
                                                         float* __restrict__ loss_b, int Bg, int Vk, int top_k,
                                                         int sim_mode, float grad_scale, float loss_scale, float* loss_sum) {
    __shared__ float red[4];
    const int row = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    float* lg = logits + (long long)row * Vk;
    const int n4 = Vk >> 2;
    f32x4 x[8];
    float mx = -INFINITY;
    if (REG) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = tid + 256 * u;
            x[u] = i < n4 ? reinterpret_cast<const f32x4*>(lg)[i] : f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
            mx = fmaxf(fmaxf(mx, fmaxf(x[u][0], x[u][1])), fmaxf(x[u][2], x[u][3]));
        }
    } else {
        for (int v = tid; v < Vk; v += blockDim.x) mx = fmaxf(mx, lg[v]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if (lane == 0) red[wv] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float se = 0.f;
    if (REG) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                x[u][k] = expf(x[u][k] - mx);  // exp(-inf) = 0 in the padding slots
                se += x[u][k];
            }
    } else {
        for (int v = tid; v < Vk; v += blockDim.x) se += expf(lg[v] - mx);
    }
    se = wave_sum(se);
    if (lane == 0) red[wv] = se;
    __syncthreads();
    se = red[0] + red[1] + red[2] + red[3];
    const float lse = mx + logf(se);
    // targets: normalised mask, assignment semantics -> slot j is live iff no later slot repeats its index.  The row's
    // top_k (index, mask) pairs are staged in LDS and the slots are worked on in parallel.
    extern __shared__ __align__(16) char kw_smem[];
    int64_t* s_idx = reinterpret_cast<int64_t*>(kw_smem);
    float* s_t = reinterpret_cast<float*>(s_idx + top_k);  // mask, then target weight t_j (0 for dead slots)
    float* s_l = s_t + top_k;                               // -(log_softmax at idx_j) * t_j
    for (int j = tid; j < top_k; j += blockDim.x) {
        s_idx[j] = kw_idx[(long long)row * top_k + j];
        s_t[j] = kw_mask[(long long)row * top_k + j];
    }
    __syncthreads();
    float msum = 0.f;
    for (int j = 0; j < top_k; ++j) msum += s_t[j];
    const float inv = 1.f / (msum + 1e-7f);
    __syncthreads();
    for (int j = tid; j < top_k; j += blockDim.x) {
        const int64_t id = s_idx[j];
        bool live = true;
        for (int j2 = j + 1; j2 < top_k; ++j2) live = live && (s_idx[j2] != id);
        const float t = live ? s_t[j] * inv : 0.f;
        s_l[j] = live ? -(lg[id] - lse) * t : 0.f;
        s_t[j] = t;  // only slot j's owner reads the mask s_t[j] after the barrier above
    }
    __syncthreads();
    float tsum = 0.f;
    for (int j = 0; j < top_k; ++j) tsum += s_t[j];
    float lsum = 0.f;
    if (tid == 0) {
        for (int j = 0; j < top_k; ++j) lsum += s_l[j];
        loss_b[row] = lsum;
    }
    __syncthreads();  // every lg[id] above is read before anybody overwrites the row
    // sim_mode (train_gan.py:288-296): keywords_weight = sim.masked_fill(sim != 1, 0); the row's gradient weight is
    // keywords_weight * grad_scale and *loss_sum += loss_b * keywords_weight * loss_scale
    float wb = w ? w[row] : 1.f;
    if (sim_mode) {
        const float kww = (w && w[row] != 1.f) ? 0.f : 1.f;
        wb = kww * grad_scale;
        if (tid == 0 && loss_sum) atomicAdd(loss_sum, lsum * kww * loss_scale);
    }
    if (REG) {
        const float sc = wb * tsum / se;  // softmax = exp(x - mx) / se
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = tid + 256 * u;
            if (i < n4) reinterpret_cast<f32x4*>(lg)[i] = x[u] * sc;
        }
    } else {
        for (int v = tid; v < Vk; v += blockDim.x) lg[v] = wb * expf(lg[v] - lse) * tsum;
    }
    __syncthreads();
    for (int j = tid; j < top_k; j += blockDim.x)
        if (s_t[j] != 0.f) lg[s_idx[j]] -= wb * s_t[j];  // live slots have distinct indices
}

// ---- whole-iteration kernels of the WGAN-GP step engine (gzsl_text_amd/gan_engine.py) --------------------------------
// A lane owns V adjacent hidden columns and walks the rows of its row block (rows interleaved over the block's four
// waves), so the per-column sums (d fc2.weight) stay in registers until one atomic per column and row block; everything
// per-row-and-column is a pure map.  V = 4 when H % 4 == 0 and the buffers are 16-byte aligned.
template <int V>
struct VecF {
    float v[V];
};
template <int V>
__device__ __forceinline__ VecF<V> ldv(const float* p) {
    VecF<V> r;
    if (V == 4) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(p);
#pragma unroll
        for (int i = 0; i < V; ++i) r.v[i] = t[i];
    } else {
#pragma unroll
        for (int i = 0; i < V; ++i) r.v[i] = p[i];
    }
    return r;
}
template <int V>
__device__ __forceinline__ void stv(float* p, const VecF<V>& r) {
    if (V == 4) {
        *reinterpret_cast<f32x4*>(p) = f32x4{r.v[0], r.v[1 % V], r.v[2 % V], r.v[3 % V]};
    } else {
#pragma unroll
        for (int i = 0; i < V; ++i) p[i] = r.v[i];
    }
}

// Rows per row block and waves per workgroup: 16 rows x 4 waves for small batches; 64 rows x 16 waves from 1024 rows on
// (same rows per wave, a quarter of the per-column atomics: 4096 rows x 800 columns would otherwise queue 256 atomics on
// every column).
#define HEAD_MAXW 16
static inline int head_rb(int Bg) { return Bg >= 1024 ? 64 : 16; }
static inline int head_threads(int Bg) { return Bg >= 1024 ? 1024 : 256; }

// Critic iteration, everything between the fc1 GEMMs and the gradient GEMMs (train_gan.py:244-256 +
// generative_module.py:22-38, closed form):
//   hf [2Bg,H] = [real; fake] @ W1[:, :D]^T,  lh [Bg,H] = lab @ W1[:, D:]^T + b1   (pre-activations h = hf + lh)
//   wb = sim_b / Bg (1 / Bg without sim);  s(x) = LeakyReLU'(x) in {1, 0.2}
//   lst rows [0,Bg)    = d(-critic_real)/dh_real = -wb * w2 * s(h_real)
//       rows [Bg,2Bg)  = d(+critic_fake)/dh_fake = +wb * w2 * s(h_fake)
//       rows [2Bg,3Bg) = v = s(h_hat) * w2,  h_hat = alpha h_real + (1 - alpha) h_fake   (fc1 is affine: no third GEMM)
//   dhs = lst[real] + lst[fake] (label half / bias gradient), db1 += column sums of dhs (d fc1.bias), sl = s(h_hat)
//   dw2 += sum_b wb (lrelu(h_fake) - lrelu(h_real));  d fc2.bias is exactly -sum wb + sum wb = 0 (the reference's
//   autograd gives 0.0 too) and is left untouched;  cost += critic_fake - critic_real (fc2.bias cancels).
template <int V>
__global__ __launch_bounds__(1024) void critic_head_kernel(const float* __restrict__ hf, const float* __restrict__ lh,
                                                          const float* __restrict__ w2, const float* __restrict__ sim,
                                                          const float* __restrict__ alpha, float* __restrict__ lst,
                                                          float* __restrict__ dhs, float* __restrict__ sl, float* dw2,
                                                           float* db1, float* cost, int Bg, int H, int rb) {
    __shared__ float red[HEAD_MAXW][64 * V];
    __shared__ float redc[HEAD_MAXW];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int col = (blockIdx.x * 64 + lane) * V;
    const bool live = col < H;
    const int r0 = blockIdx.y * rb, r1 = min(Bg, r0 + rb), nw = blockDim.x >> 6;
    const float inv_bg = 1.f / (float)Bg;
    VecF<V> w, acc, accb;
#pragma unroll
    for (int i = 0; i < V; ++i) acc.v[i] = 0.f, accb.v[i] = 0.f, w.v[i] = 0.f;
    if (live) w = ldv<V>(w2 + col);
    float c = 0.f;
    if (live) {
        for (int r = r0 + wv; r < r1; r += nw) {
            const float wb = (sim ? sim[r] : 1.f) * inv_bg, a = alpha[r];
            const long long o = (long long)r * H + col;
            const VecF<V> l = ldv<V>(lh + o), fr = ldv<V>(hf + o), fk = ldv<V>(hf + (long long)Bg * H + o);
            VecF<V> dr, dk, vv, ds, ss;
#pragma unroll
            for (int i = 0; i < V; ++i) {
                const float hr = fr.v[i] + l.v[i], hk = fk.v[i] + l.v[i];
                const float sr = hr > 0.f ? 1.f : 0.2f, sk = hk > 0.f ? 1.f : 0.2f;
                const float hh = a * hr + (1.f - a) * hk;
                const float sh = hh > 0.f ? 1.f : 0.2f;
                dr.v[i] = -wb * w.v[i] * sr;
                dk.v[i] = wb * w.v[i] * sk;
                vv.v[i] = sh * w.v[i];
                ds.v[i] = dr.v[i] + dk.v[i];
                ss.v[i] = sh;
                const float diff = (hk > 0.f ? hk : 0.2f * hk) - (hr > 0.f ? hr : 0.2f * hr);
                acc.v[i] += wb * diff;
                accb.v[i] += ds.v[i];
                c += wb * diff * w.v[i];
            }
            stv<V>(lst + o, dr);
            stv<V>(lst + (long long)Bg * H + o, dk);
            stv<V>(lst + 2ll * Bg * H + o, vv);
            stv<V>(dhs + o, ds);
            stv<V>(sl + o, ss);
        }
    }
#pragma unroll
    for (int i = 0; i < V; ++i) red[wv][lane * V + i] = acc.v[i];
    c = wave_sum(c);
    if (lane == 0) redc[wv] = c;
    __syncthreads();
    if (wv == 0 && live) {
#pragma unroll
        for (int i = 0; i < V; ++i) {
            const int k = lane * V + i;
            float t = 0.f;
            for (int u = 0; u < nw; ++u) t += red[u][k];
            atomicAdd(dw2 + col + i, t);
        }
    }
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int u = 0; u < nw; ++u) t += redc[u];
        atomicAdd(cost, t);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < V; ++i) red[wv][lane * V + i] = accb.v[i];
    __syncthreads();
    if (wv == 0 && live) {
#pragma unroll
        for (int i = 0; i < V; ++i) {
            const int k = lane * V + i;
            float t = 0.f;
            for (int u = 0; u < nw; ++u) t += red[u][k];
            atomicAdd(db1 + col + i, t);
        }
    }
}

// Generator iteration (train_gan.py:299-306): the critic is frozen, only d(-critic_fake)/dh is needed.
//   h = hf + lh;  dh = -wb * w2 * s(h);  cost += -sum_b wb (lrelu(h_b) . w2 + b2)
template <int V>
__global__ __launch_bounds__(1024) void gen_head_kernel(const float* __restrict__ hf, const float* __restrict__ lh,
                                                       const float* __restrict__ w2, const float* __restrict__ b2,
                                                       const float* __restrict__ sim, float* __restrict__ dh, float* cost,
                                                       int Bg, int H, int rb) {
    __shared__ float redc[HEAD_MAXW];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int col = (blockIdx.x * 64 + lane) * V;
    const bool live = col < H;
    const int r0 = blockIdx.y * rb, r1 = min(Bg, r0 + rb), nw = blockDim.x >> 6;
    const float inv_bg = 1.f / (float)Bg;
    VecF<V> w;
#pragma unroll
    for (int i = 0; i < V; ++i) w.v[i] = 0.f;
    if (live) w = ldv<V>(w2 + col);
    float c = 0.f;
    for (int r = r0 + wv; r < r1; r += nw) {
        const float wb = (sim ? sim[r] : 1.f) * inv_bg;
        if (blockIdx.x == 0 && lane == 0) c -= wb * b2[0];
        if (!live) continue;
        const long long o = (long long)r * H + col;
        const VecF<V> l = ldv<V>(lh + o), f = ldv<V>(hf + o);
        VecF<V> d;
#pragma unroll
        for (int i = 0; i < V; ++i) {
            const float h = f.v[i] + l.v[i];
            d.v[i] = -wb * w.v[i] * (h > 0.f ? 1.f : 0.2f);
            c -= wb * w.v[i] * (h > 0.f ? h : 0.2f * h);
        }
        stv<V>(dh + o, d);
    }
    c = wave_sum(c);
    if (lane == 0) redc[wv] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int u = 0; u < nw; ++u) t += redc[u];
        atomicAdd(cost, t);
    }
}

// dw2[j] += sum_b dv[b,j] * s[b,j] with the column-owner layout of the head kernels
template <int V>
__global__ __launch_bounds__(1024) void gp_dw2_cols_kernel(const float* __restrict__ dv, const float* __restrict__ s,
                                                          float* dw2, int Bg, int H, int rb) {
    __shared__ float red[HEAD_MAXW][64 * V];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int col = (blockIdx.x * 64 + lane) * V;
    const bool live = col < H;
    const int r0 = blockIdx.y * rb, r1 = min(Bg, r0 + rb), nw = blockDim.x >> 6;
    VecF<V> acc;
#pragma unroll
    for (int i = 0; i < V; ++i) acc.v[i] = 0.f;
    if (live) {
        for (int r = r0 + wv; r < r1; r += nw) {
            const long long o = (long long)r * H + col;
            const VecF<V> a = ldv<V>(dv + o), b = ldv<V>(s + o);
#pragma unroll
            for (int i = 0; i < V; ++i) acc.v[i] += a.v[i] * b.v[i];
        }
    }
#pragma unroll
    for (int i = 0; i < V; ++i) red[wv][lane * V + i] = acc.v[i];
    __syncthreads();
    if (wv == 0 && live) {
#pragma unroll
        for (int i = 0; i < V; ++i) {
            const int k = lane * V + i;
            float t = 0.f;
            for (int u = 0; u < nw; ++u) t += red[u][k];
            atomicAdd(dw2 + col + i, t);
        }
    }
}

// torch.optim.Adam (no weight decay, no amsgrad) over a FLAT parameter buffer; the step counter lives on the device so
// that a recorded HIP graph replays correctly.  Every workgroup keeps its OWN copy of the count in state[blockIdx.x]
// (the grid of a given buffer never changes), so there is no cross-workgroup ordering to establish: no atomics, no fences.
//   m = m + (g - m)(1 - b1);  v = b2 v + (1 - b2) g^2;  p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
#define ADAM_MAX_BLOCKS 1024
__global__ __launch_bounds__(256) void adam_flat_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                        float* __restrict__ m, float* __restrict__ v, long long n, float lr,
                                                        float b1, float b2, float eps, int* state) {
    __shared__ float sh[2];
    if (threadIdx.x == 0) {
        const int t = state[blockIdx.x] + 1;
        state[blockIdx.x] = t;
        const double bc1 = 1.0 - pow((double)b1, (double)t), bc2 = 1.0 - pow((double)b2, (double)t);
        sh[0] = (float)((double)lr / bc1);
        sh[1] = (float)sqrt(bc2);
    }
    __syncthreads();
    const float step_size = sh[0], bc2_sqrt = sh[1];
    const long long n4 = n >> 2;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        f32x4 pp = reinterpret_cast<f32x4*>(p)[i], mm = reinterpret_cast<f32x4*>(m)[i], vv = reinterpret_cast<f32x4*>(v)[i];
        const f32x4 gg = reinterpret_cast<const f32x4*>(g)[i];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            mm[k] = mm[k] + (gg[k] - mm[k]) * (1.f - b1);
            vv[k] = vv[k] * b2 + (1.f - b2) * gg[k] * gg[k];
            pp[k] -= step_size * (mm[k] / (sqrtf(vv[k]) / bc2_sqrt + eps));
        }
        reinterpret_cast<f32x4*>(p)[i] = pp;
        reinterpret_cast<f32x4*>(m)[i] = mm;
        reinterpret_cast<f32x4*>(v)[i] = vv;
    }
    if (blockIdx.x == 0)
        for (long long i = (n4 << 2) + threadIdx.x; i < n; i += blockDim.x) {
            const float gg = g[i];
            const float mm = m[i] + (gg - m[i]) * (1.f - b1);
            const float vv = v[i] * b2 + (1.f - b2) * gg * gg;
            m[i] = mm, v[i] = vv;
            p[i] -= step_size * (mm / (sqrtf(vv) / bc2_sqrt + eps));
        }
}

extern "C" int gzsl_gan_critic_head(const float* hf, const float* lh, const float* w2, const float* sim,
                                    const float* alpha, float* lst, float* dhs, float* sl, float* dw2, float* db1,
                                    float* cost, int Bg, int H, void* stream) {
    if (Bg <= 0 || H <= 0) return 0;
    const bool v4 = (H & 3) == 0 && al16(hf) && al16(lh) && al16(w2) && al16(lst) && al16(dhs) && al16(sl);
    const int rb = head_rb(Bg);
    const dim3 grid(cdiv(v4 ? H / 4 : H, 64), cdiv(Bg, rb)), block(head_threads(Bg));
    if (v4)
        hipLaunchKernelGGL(critic_head_kernel<4>, grid, block, 0, ST, hf, lh, w2, sim, alpha, lst, dhs, sl, dw2, db1, cost, Bg, H, rb);
    else
        hipLaunchKernelGGL(critic_head_kernel<1>, grid, block, 0, ST, hf, lh, w2, sim, alpha, lst, dhs, sl, dw2, db1, cost, Bg, H, rb);
    GZSL_CHECK_LAUNCH();
    return 0;
}

extern "C" int gzsl_gan_gen_head(const float* hf, const float* lh, const float* w2, const float* b2, const float* sim,
                                 float* dh, float* cost, int Bg, int H, void* stream) {
    if (Bg <= 0 || H <= 0) return 0;
    const bool v4 = (H & 3) == 0 && al16(hf) && al16(lh) && al16(w2) && al16(dh);
    const int rb = head_rb(Bg);
    const dim3 grid(cdiv(v4 ? H / 4 : H, 64), cdiv(Bg, rb)), block(head_threads(Bg));
    if (v4) hipLaunchKernelGGL(gen_head_kernel<4>, grid, block, 0, ST, hf, lh, w2, b2, sim, dh, cost, Bg, H, rb);
    else hipLaunchKernelGGL(gen_head_kernel<1>, grid, block, 0, ST, hf, lh, w2, b2, sim, dh, cost, Bg, H, rb);
    GZSL_CHECK_LAUNCH();
    return 0;
}

extern "C" int gzsl_adam_flat_step(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1,
                                   float beta2, float eps, int* state, void* stream) {
    if (n <= 0) return 0;
    if (!al16(p) || !al16(g) || !al16(m) || !al16(v)) return GZSL_EINVAL;
    long long blocks = (n / 4 + 255) / 256;  // a function of n alone: the per-workgroup counters rely on it
    if (blocks > ADAM_MAX_BLOCKS) blocks = ADAM_MAX_BLOCKS;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(adam_flat_kernel, dim3((int)blocks), dim3(256), 0, ST, p, g, m, v, n, lr, beta1, beta2, eps, state);
    GZSL_CHECK_LAUNCH();
    return 0;
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
    const dim3 grid(min(cdiv(Bg, 4), 2048)), block(256);
    if ((D & 3) == 0 && al16(g)) hipLaunchKernelGGL(gp_rownorm_kernel<4>, grid, block, 0, ST, g, gp, Bg, D, lam);
    else hipLaunchKernelGGL(gp_rownorm_kernel<1>, grid, block, 0, ST, g, gp, Bg, D, lam);
    GZSL_CHECK_LAUNCH();
    return 0;
}

extern "C" int gzsl_gp_dw2(const float* dv, const float* s, float* dw2, int Bg, int H, void* stream) {
    if (Bg <= 0) return 0;
    const bool v4 = (H & 3) == 0 && al16(dv) && al16(s);
    const int rb = head_rb(Bg);
    const dim3 grid(cdiv(v4 ? H / 4 : H, 64), cdiv(Bg, rb)), block(head_threads(Bg));
    if (v4) hipLaunchKernelGGL(gp_dw2_cols_kernel<4>, grid, block, 0, ST, dv, s, dw2, Bg, H, rb);
    else hipLaunchKernelGGL(gp_dw2_cols_kernel<1>, grid, block, 0, ST, dv, s, dw2, Bg, H, rb);
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
    const bool v4 = (H & 3) == 0 && al16(act) && (!da || al16(da));
    const int rb = head_rb(Bg);
    const dim3 grid(cdiv(v4 ? H / 4 : H, 64), cdiv(Bg, rb)), block(head_threads(Bg));
    if (v4) hipLaunchKernelGGL(rowdot_bwd_kernel<4>, grid, block, 0, ST, dout, w, act, da, dw, db, Bg, H, rb, lrelu_in);
    else hipLaunchKernelGGL(rowdot_bwd_kernel<1>, grid, block, 0, ST, dout, w, act, da, dw, db, Bg, H, rb, lrelu_in);
    GZSL_CHECK_LAUNCH();
    return 0;
}

static int keyword_ce_launch(float* logits, const int64_t* kw_idx, const float* kw_mask, const float* w, float* loss_b,
                             int Bg, int Vk, int top_k, int sim_mode, float grad_scale, float loss_scale, float* loss_sum,
                             void* stream) {
    if (Bg <= 0) return 0;
    if (Vk <= 0 || top_k <= 0 || top_k > 2048) return GZSL_EINVAL;
    const size_t smem = (size_t)top_k * 16;
    if ((Vk & 3) == 0 && Vk <= 8192 && al16(logits))
        hipLaunchKernelGGL(keyword_ce_kernel<true>, dim3(Bg), dim3(256), smem, ST, logits, kw_idx, kw_mask, w, loss_b, Bg, Vk,
                           top_k, sim_mode, grad_scale, loss_scale, loss_sum);
    else
        hipLaunchKernelGGL(keyword_ce_kernel<false>, dim3(Bg), dim3(256), smem, ST, logits, kw_idx, kw_mask, w, loss_b, Bg, Vk,
                           top_k, sim_mode, grad_scale, loss_scale, loss_sum);
    GZSL_CHECK_LAUNCH();
    return 0;
}

extern "C" int gzsl_keyword_ce(float* logits, const int64_t* kw_idx, const float* kw_mask, const float* w,
                               float* loss_b, int Bg, int Vk, int top_k, void* stream) {
    return keyword_ce_launch(logits, kw_idx, kw_mask, w, loss_b, Bg, Vk, top_k, 0, 1.f, 1.f, nullptr, stream);
}

extern "C" int gzsl_keyword_ce_sim(float* logits, const int64_t* kw_idx, const float* kw_mask, const float* sim,
                                   float grad_scale, float loss_scale, float* loss_b, float* loss_sum, int Bg, int Vk,
                                   int top_k, void* stream) {
    return keyword_ce_launch(logits, kw_idx, kw_mask, sim, loss_b, Bg, Vk, top_k, 1, grad_scale, loss_scale, loss_sum, stream);
}
