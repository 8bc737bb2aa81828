// HBM-bound kernels of the base feature extractor: embedding gather / scatter-add, label-description average,
// CSR neighbour mean, GRU gate math, classifier head, LDAM + BCE, AdamW, and the small backward helpers.
// Each kernel is a grid-stride loop with lanes along the contiguous (feature) dimension, so every wave touches
// 256 consecutive bytes per instruction; reductions go wave-shuffle -> one float atomic per wave/row.
#include <mutex>
#include "common.h"
#include "rng.h"
#include "../../include/gzsl_hip.h"

#define EW_THREADS 256
static inline int ew_blocks(long long n, int per = EW_THREADS) {
    long long b = (n + per - 1) / per;
    if (b > 8192) b = 8192;
    if (b < 1) b = 1;
    return (int)b;
}
#define GRID_STRIDE(i, n) \
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < (n); i += (long long)gridDim.x * blockDim.x)

// -------------------------------------------------------------------------------------------- embedding
__global__ void emb_gather_kernel(const float* __restrict__ W, const int64_t* __restrict__ ids,
                                  const float* __restrict__ mask, float* __restrict__ out, long long rows, int E,
                                  long long row_offset, uint32_t seed, uint32_t thresh, float scale) {
    const long long n = rows * E;
    GRID_STRIDE(i, n) {
        const long long row = i / E;
        const int e = (int)(i - row * E);
        float v = W[ids[row] * E + e];
        if (mask) v *= mask[row];
        if (thresh) {
            const uint32_t u = gz_u32(gz_row_key(seed, GZSL_STREAM_EMB, (uint32_t)(row + row_offset)), (uint32_t)e);
            v = (u >= thresh) ? v * scale : 0.f;
        }
        out[i] = v;
    }
}

// 16-byte form (E % 4 == 0, aligned bases): a thread owns four consecutive columns of one row; the row's id and mask are
// read once per thread (not per element), the divide is per four elements, loads / stores are dwordx4.  Bit-identical
// to the scalar kernel (same hash per element, same multiplications).
__global__ void emb_gather_v4_kernel(const float* __restrict__ W, const int64_t* __restrict__ ids,
                                     const float* __restrict__ mask, float* __restrict__ out, long long rows, int E4,
                                     long long row_offset, uint32_t seed, uint32_t thresh, float scale) {
    const long long n = rows * E4;
    const float4* W4 = reinterpret_cast<const float4*>(W);
    float4* out4 = reinterpret_cast<float4*>(out);
    GRID_STRIDE(i, n) {
        const long long row = i / E4;
        const int c = (int)(i - row * E4);
        float4 v = W4[ids[row] * E4 + c];
        if (mask) {
            const float mk = mask[row];
            v.x *= mk, v.y *= mk, v.z *= mk, v.w *= mk;
        }
        if (thresh) {
            const uint32_t key = gz_row_key(seed, GZSL_STREAM_EMB, (uint32_t)(row + row_offset));
            v.x = (gz_u32(key, 4u * c) >= thresh) ? v.x * scale : 0.f;
            v.y = (gz_u32(key, 4u * c + 1u) >= thresh) ? v.y * scale : 0.f;
            v.z = (gz_u32(key, 4u * c + 2u) >= thresh) ? v.z * scale : 0.f;
            v.w = (gz_u32(key, 4u * c + 3u) >= thresh) ? v.w * scale : 0.f;
        }
        out4[i] = v;
    }
}

__global__ void emb_scatter_kernel(float* __restrict__ dW, const int64_t* __restrict__ ids,
                                   const float* __restrict__ mask, const float* __restrict__ dOut, long long rows, int E,
                                   int64_t pad_idx, long long row_offset, uint32_t seed, uint32_t thresh, float scale) {
    const long long n = rows * E;
    GRID_STRIDE(i, n) {
        const long long row = i / E;
        const int e = (int)(i - row * E);
        const int64_t id = ids[row];
        if (id == pad_idx) continue;
        float v = dOut[i];
        if (mask) v *= mask[row];
        if (thresh) {
            const uint32_t u = gz_u32(gz_row_key(seed, GZSL_STREAM_EMB, (uint32_t)(row + row_offset)), (uint32_t)e);
            v = (u >= thresh) ? v * scale : 0.f;
        }
        if (v != 0.f) atomicAdd(dW + id * E + e, v);
    }
}

// ---- scatter-add grouped by id inside 128-row chunks ------------------------------------------------------------------
// Word ids follow a Zipf law: thousands of rows of a note batch add into the same few embedding rows, and same-address
// float atomics serialise at the memory side (MI355X_MICROARCH.md: "every workgroup into ONE row: 14x slower").  Each
// workgroup sorts the ids of its 128 consecutive rows in LDS (bitonic), stages the rows' gradients in sorted order and
// sums runs of equal ids down the columns: one atomic per (run, column) instead of one per (row, column).
#define ES_CHUNK 128
__global__ __launch_bounds__(256) void emb_scatter_grouped_kernel(
    float* __restrict__ dW, const int64_t* __restrict__ ids, const float* __restrict__ mask,
    const float* __restrict__ dOut, long long rows, int E, int64_t pad_idx, long long row_offset, uint32_t seed,
    uint32_t thresh, float scale) {
    extern __shared__ __attribute__((aligned(16))) char es_smem[];
    unsigned long long* key = reinterpret_cast<unsigned long long*>(es_smem);  // [128] (id << 8 | row in chunk)
    float* tile = reinterpret_cast<float*>(key + ES_CHUNK);                      // [128][E], rows in sorted order
    const int tid = threadIdx.x;
    const long long r0 = (long long)blockIdx.x * ES_CHUNK;
    if (tid < ES_CHUNK) {
        const long long r = r0 + tid;
        unsigned long long k = ~0ull;  // rows past the end and padding ids sort last and are skipped
        if (r < rows) {
            const int64_t id = ids[r];
            if (id != pad_idx && id >= 0) k = ((unsigned long long)id << 8) | (unsigned)tid;
        }
        key[tid] = k;
    }
    __syncthreads();
    for (int size = 2; size <= ES_CHUNK; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            if (tid < ES_CHUNK) {
                const int j = tid ^ stride;
                if (j > tid) {
                    const unsigned long long a = key[tid], b = key[j];
                    const bool up = (tid & size) == 0;
                    if ((a > b) == up) {
                        key[tid] = b;
                        key[j] = a;
                    }
                }
            }
            __syncthreads();
        }
    }
    for (int idx = tid; idx < ES_CHUNK * E; idx += 256) {
        const int j = idx / E, e = idx - j * E;
        const unsigned long long k = key[j];
        float v = 0.f;
        if (k != ~0ull) {
            const long long r = r0 + (long long)(k & 255);
            v = dOut[r * E + e];
            if (mask) v *= mask[r];
            if (thresh) {
                const uint32_t u = gz_u32(gz_row_key(seed, GZSL_STREAM_EMB, (uint32_t)(r + row_offset)), (uint32_t)e);
                v = (u >= thresh) ? v * scale : 0.f;
            }
        }
        tile[idx] = v;
    }
    __syncthreads();
    // thread (half, e): sums runs of equal ids over 64 sorted rows of column e
    const int halves = E <= 128 ? 2 : 1, per = ES_CHUNK / halves;
    const int half = E <= 128 ? tid >> 7 : 0;
    for (int e = E <= 128 ? (tid & 127) : tid; e < E; e += E <= 128 ? 128 : 256) {
        float acc = 0.f;
        unsigned long long cur = key[half * per] >> 8;
        for (int j = half * per; j < (half + 1) * per; ++j) {
            const unsigned long long k = key[j];
            if (k == ~0ull) break;
            const unsigned long long id = k >> 8;
            if (id != cur) {
                if (acc != 0.f) atomicAdd(dW + (long long)cur * E + e, acc);
                acc = 0.f;
                cur = id;
            }
            acc += tile[j * E + e];
        }
        if (acc != 0.f) atomicAdd(dW + (long long)cur * E + e, acc);
    }
}

__global__ void label_avg_fwd_kernel(const float* __restrict__ W, const int64_t* __restrict__ idx,
                                     const float* __restrict__ mask, float* __restrict__ out, int N, int Td, int E) {
    const long long n = (long long)N * E;
    GRID_STRIDE(i, n) {
        const int row = (int)(i / E), e = (int)(i % E);
        float s = 0.f, cnt = 0.f;
        for (int j = 0; j < Td; ++j) {
            const float mk = mask[(long long)row * Td + j];
            cnt += mk;
            s += mk * W[idx[(long long)row * Td + j] * E + e];
        }
        out[i] = s / cnt;
    }
}

// 16-byte form: a thread owns four columns of one label row and gathers its (at most Td) description-word rows as
// dwordx4, eight loads in flight; same summation order over j as the scalar kernel (bit-identical).
__global__ void label_avg_fwd_v4_kernel(const float* __restrict__ W, const int64_t* __restrict__ idx,
                                        const float* __restrict__ mask, float* __restrict__ out, int N, int Td, int E4) {
    const long long n = (long long)N * E4;
    const float4* W4 = reinterpret_cast<const float4*>(W);
    GRID_STRIDE(i, n) {
        const int row = (int)(i / E4), c = (int)(i - (long long)row * E4);
        const int64_t* ip = idx + (long long)row * Td;
        const float* mp = mask + (long long)row * Td;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        float cnt = 0.f;
        int j = 0;
        for (; j + 8 <= Td; j += 8) {
            float4 t[8];
            float mk[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                mk[u] = mp[j + u];
                t[u] = W4[ip[j + u] * E4 + c];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                cnt += mk[u];
                s.x += mk[u] * t[u].x, s.y += mk[u] * t[u].y, s.z += mk[u] * t[u].z, s.w += mk[u] * t[u].w;
            }
        }
        for (; j < Td; ++j) {
            const float mk = mp[j];
            const float4 t = W4[ip[j] * E4 + c];
            cnt += mk;
            s.x += mk * t.x, s.y += mk * t.y, s.z += mk * t.z, s.w += mk * t.w;
        }
        reinterpret_cast<float4*>(out)[i] = make_float4(s.x / cnt, s.y / cnt, s.z / cnt, s.w / cnt);
    }
}

__global__ void label_avg_bwd_kernel(float* __restrict__ dW, const float* __restrict__ davg,
                                     const int64_t* __restrict__ idx, const float* __restrict__ mask, int N, int Td,
                                     int E, int64_t pad_idx) {
    const long long n = (long long)N * E;
    GRID_STRIDE(i, n) {
        const int row = (int)(i / E), e = (int)(i % E);
        float cnt = 0.f;
        for (int j = 0; j < Td; ++j) cnt += mask[(long long)row * Td + j];
        const float g = davg[i] / cnt;
        for (int j = 0; j < Td; ++j) {
            const int64_t id = idx[(long long)row * Td + j];
            const float mk = mask[(long long)row * Td + j];
            if (id != pad_idx && mk != 0.f) atomicAdd(dW + id * E + e, mk * g);
        }
    }
}

__global__ void spmm_csr_kernel(const int64_t* __restrict__ rowptr, const int64_t* __restrict__ col,
                                const float* __restrict__ val, const float* __restrict__ x,
                                const float* __restrict__ pre, const float* __restrict__ post, float* __restrict__ out,
                                int N, int H, int accumulate) {
    const long long n = (long long)N * H;
    GRID_STRIDE(i, n) {
        const int row = (int)(i / H), h = (int)(i % H);
        float s = 0.f;
        for (int64_t p = rowptr[row]; p < rowptr[row + 1]; ++p) {
            const int64_t c = col[p];
            float w = val[p];
            if (pre) w *= pre[c];
            s += w * x[c * H + h];
        }
        if (post) s *= post[row];
        if (accumulate == 2) {
            if (rowptr[row] != rowptr[row + 1]) atomicAdd(out + i, s);
        } else {
            out[i] = accumulate ? out[i] + s : s;
        }
    }
}

// 16-byte form (H % 4 == 0).  accumulate: 0 store, 1 add (plain read-modify-write), 2 add with float atomics and skip
// empty rows (the label-description adjoint adds into the word-table gradient while the note encoder's scatter-add,
// on another stream, adds into the same table).
__global__ void spmm_csr_v4_kernel(const int64_t* __restrict__ rowptr, const int64_t* __restrict__ col,
                                   const float* __restrict__ val, const float* __restrict__ x,
                                   const float* __restrict__ pre, const float* __restrict__ post, float* __restrict__ out,
                                   int N, int H4, int accumulate) {
    const long long n = (long long)N * H4;
    const float4* x4 = reinterpret_cast<const float4*>(x);
    GRID_STRIDE(i, n) {
        const int row = (int)(i / H4), h = (int)(i - (long long)row * H4);
        const int64_t p0 = rowptr[row], p1 = rowptr[row + 1];
        if (accumulate == 2 && p0 == p1) continue;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int64_t p = p0; p < p1; ++p) {
            const int64_t c = col[p];
            float w = val[p];
            if (pre) w *= pre[c];
            const float4 v = x4[c * H4 + h];
            s.x += w * v.x, s.y += w * v.y, s.z += w * v.z, s.w += w * v.w;
        }
        if (post) {
            const float q = post[row];
            s.x *= q, s.y *= q, s.z *= q, s.w *= q;
        }
        float* o = out + 4 * i;
        if (accumulate == 2) {
            atomicAdd(o, s.x), atomicAdd(o + 1, s.y), atomicAdd(o + 2, s.z), atomicAdd(o + 3, s.w);
        } else {
            if (accumulate) {
                const float4 t = *reinterpret_cast<const float4*>(o);
                s.x += t.x, s.y += t.y, s.z += t.z, s.w += t.w;
            }
            *reinterpret_cast<float4*>(o) = s;
        }
    }
}

// -------------------------------------------------------------------------------------------- GRU gates
// 16-byte forms (H % 4 == 0): a thread owns four columns of one row; same arithmetic per element as the scalar kernels.
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

__global__ void gru_gate_fwd_v4_kernel(const float* __restrict__ gi, const float* __restrict__ gh,
                                       const float* __restrict__ h, float* __restrict__ hn, int N, int H) {
    const int H4 = H >> 2;
    const long long n = (long long)N * H4;
    GRID_STRIDE(i, n) {
        const long long row = i / H4;
        const int c = 4 * (int)(i - row * H4);
        const float* a = gi + row * 3 * H + c;
        const float* b = gh + row * 3 * H + c;
        const float4 ar = ld4(a), az = ld4(a + H), an = ld4(a + 2 * H), br = ld4(b), bz = ld4(b + H), bn = ld4(b + 2 * H);
        const float4 hh = ld4(h + row * H + c);
        float4 o;
#define GZ_GATE(q)                                             \
        {                                                      \
            const float r = sigmoidf_(ar.q + br.q);            \
            const float z = sigmoidf_(az.q + bz.q);            \
            const float nn = tanhf(an.q + r * bn.q);           \
            o.q = (1.f - z) * nn + z * hh.q;                   \
        }
        GZ_GATE(x) GZ_GATE(y) GZ_GATE(z) GZ_GATE(w)
#undef GZ_GATE
        st4(hn + row * H + c, o);
    }
}

__global__ void gru_gate_bwd_v4_kernel(const float* __restrict__ dhn, const float* __restrict__ gi,
                                       const float* __restrict__ gh, const float* __restrict__ h,
                                       float* __restrict__ dgi, float* __restrict__ dgh, float* __restrict__ dh, int N,
                                       int H) {
    const int H4 = H >> 2;
    const long long n = (long long)N * H4;
    GRID_STRIDE(i, n) {
        const long long row = i / H4;
        const int c = 4 * (int)(i - row * H4);
        const float* a = gi + row * 3 * H + c;
        const float* b = gh + row * 3 * H + c;
        const float4 ar = ld4(a), az = ld4(a + H), an = ld4(a + 2 * H), br = ld4(b), bz = ld4(b + H), bn = ld4(b + 2 * H);
        const float4 hh = ld4(h + row * H + c), dd = ld4(dhn + row * H + c);
        float4 o_r, o_z, o_n, o_nr, o_h;
#define GZ_GATE(q)                                                     \
        {                                                              \
            const float r = sigmoidf_(ar.q + br.q);                    \
            const float z = sigmoidf_(az.q + bz.q);                    \
            const float ghn = bn.q;                                    \
            const float nn = tanhf(an.q + r * ghn);                    \
            const float d = dd.q;                                      \
            const float dn_pre = d * (1.f - z) * (1.f - nn * nn);      \
            o_z.q = d * (hh.q - nn) * z * (1.f - z);                   \
            o_r.q = dn_pre * ghn * r * (1.f - r);                      \
            o_n.q = dn_pre;                                            \
            o_nr.q = dn_pre * r;                                       \
            o_h.q = d * z;                                             \
        }
        GZ_GATE(x) GZ_GATE(y) GZ_GATE(z) GZ_GATE(w)
#undef GZ_GATE
        float* oa = dgi + row * 3 * H + c;
        float* ob = dgh + row * 3 * H + c;
        st4(oa, o_r), st4(oa + H, o_z), st4(oa + 2 * H, o_n);
        st4(ob, o_r), st4(ob + H, o_z), st4(ob + 2 * H, o_nr);
        st4(dh + row * H + c, o_h);
    }
}

__global__ void gru_gate_fwd_kernel(const float* __restrict__ gi, const float* __restrict__ gh,
                                    const float* __restrict__ h, float* __restrict__ hn, int N, int H) {
    const long long n = (long long)N * H;
    GRID_STRIDE(i, n) {
        const long long row = i / H;
        const int c = (int)(i - row * H);
        const float* a = gi + row * 3 * H;
        const float* b = gh + row * 3 * H;
        const float r = sigmoidf_(a[c] + b[c]);
        const float z = sigmoidf_(a[H + c] + b[H + c]);
        const float nn = tanhf(a[2 * H + c] + r * b[2 * H + c]);
        hn[i] = (1.f - z) * nn + z * h[i];
    }
}

// One time step of the packed label GRU (lm_module.py:68-88, training=False): rows whose description is longer than t
// take the GRUCell update, shorter rows keep their state (pack_padded_sequence semantics).  The pooled output is
// maintained on the fly: max over the steps' hidden states EXCLUDING exact zeros (the reference's
// masked_fill(outputs == 0, -inf)), or their sum (divided by the length at the last step) for 'mean'.
// gi = x_t.W_ih^T + b_ih for step t is row b of a [Bg, T, 3H] tensor: gi + b * gi_stride.
__global__ void gru_seq_step_kernel(const float* __restrict__ gi, long long gi_stride, const float* __restrict__ gh,
                                    float* __restrict__ h, const float* __restrict__ mask, int T, int t,
                                    float* __restrict__ pool, int pool_mode, int Bg, int H) {
    const long long n = (long long)Bg * H;
    GRID_STRIDE(i, n) {
        const long long row = i / H;
        const int c = (int)(i - row * H);
        float len = 0.f;
        for (int j = 0; j < T; ++j) len += mask[row * T + j];
        if ((float)t < len) {
            const float* a = gi + row * gi_stride;
            const float* b = gh + row * 3 * H;
            const float r = sigmoidf_(a[c] + b[c]);
            const float z = sigmoidf_(a[H + c] + b[H + c]);
            const float nn = tanhf(a[2 * H + c] + r * b[2 * H + c]);
            const float hn = (1.f - z) * nn + z * h[i];
            h[i] = hn;
            if (pool_mode == 1) {
                if (hn != 0.f) pool[i] = fmaxf(pool[i], hn);
            } else if (pool_mode == 2) {
                pool[i] += hn;
            }
        }
        if (pool_mode == 2 && t == T - 1) pool[i] /= len;
    }
}

__global__ void gru_gate_bwd_kernel(const float* __restrict__ dhn, const float* __restrict__ gi,
                                    const float* __restrict__ gh, const float* __restrict__ h, float* __restrict__ dgi,
                                    float* __restrict__ dgh, float* __restrict__ dh, int N, int H) {
    const long long n = (long long)N * H;
    GRID_STRIDE(i, n) {
        const long long row = i / H;
        const int c = (int)(i - row * H);
        const float* a = gi + row * 3 * H;
        const float* b = gh + row * 3 * H;
        const float r = sigmoidf_(a[c] + b[c]);
        const float z = sigmoidf_(a[H + c] + b[H + c]);
        const float ghn = b[2 * H + c];
        const float nn = tanhf(a[2 * H + c] + r * ghn);
        const float d = dhn[i];
        const float dn_pre = d * (1.f - z) * (1.f - nn * nn);
        const float dz_pre = d * (h[i] - nn) * z * (1.f - z);
        const float dr_pre = dn_pre * ghn * r * (1.f - r);
        float* oa = dgi + row * 3 * H;
        float* ob = dgh + row * 3 * H;
        oa[c] = dr_pre;
        oa[H + c] = dz_pre;
        oa[2 * H + c] = dn_pre;
        ob[c] = dr_pre;
        ob[H + c] = dz_pre;
        ob[2 * H + c] = dn_pre * r;
        dh[i] = d * z;
    }
}

// -------------------------------------------------------------------------------------------- classifier head
__global__ void logits_fwd_kernel(const float* __restrict__ direct, const float* __restrict__ clf,
                                  float* __restrict__ logits, int B, int L, int D) {
    const int lane = threadIdx.x & 63;
    const long long rows = (long long)B * L;
    long long row = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    for (; row < rows; row += (long long)gridDim.x * (blockDim.x >> 6)) {
        const int l = (int)(row % L);
        float s = 0.f;
        for (int d = lane; d < D; d += 64) s += clf[(long long)l * D + d] * fmaxf(direct[row * D + d], 0.f);
        s = wave_sum(s);
        if (lane == 0) logits[row] = s;
    }
}

// 16-byte form (D % 4 == 0): one wave per (note, label) row, a lane owns four consecutive columns.
__global__ void logits_fwd_v4_kernel(const float* __restrict__ direct, const float* __restrict__ clf,
                                     float* __restrict__ logits, int B, int L, int D4) {
    const int lane = threadIdx.x & 63;
    const long long rows = (long long)B * L;
    long long row = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const float4* dir4 = reinterpret_cast<const float4*>(direct);
    const float4* clf4 = reinterpret_cast<const float4*>(clf);
    for (; row < rows; row += (long long)gridDim.x * (blockDim.x >> 6)) {
        const int l = (int)(row % L);
        float s = 0.f;
        for (int d = lane; d < D4; d += 64) {
            const float4 v = dir4[row * D4 + d], c = clf4[(long long)l * D4 + d];
            s += c.x * fmaxf(v.x, 0.f) + c.y * fmaxf(v.y, 0.f) + c.z * fmaxf(v.z, 0.f) + c.w * fmaxf(v.w, 0.f);
        }
        s = wave_sum(s);
        if (lane == 0) logits[row] = s;
    }
}

__global__ void logits_bwd_kernel(const float* __restrict__ direct, const float* __restrict__ clf,
                                  const float* __restrict__ dlogits, float* __restrict__ ddirect, int B, int L, int D) {
    const long long n = (long long)B * L * D;
    GRID_STRIDE(i, n) {
        const long long row = i / D;
        const int d = (int)(i - row * D);
        const int l = (int)(row % L);
        ddirect[i] = direct[i] > 0.f ? dlogits[row] * clf[(long long)l * D + d] : 0.f;
    }
}

__global__ void dclf_kernel(const float* __restrict__ direct, const float* __restrict__ dlogits,
                            float* __restrict__ dclf, int B, int L, int D, int accumulate) {
    const long long n = (long long)L * D;
    GRID_STRIDE(i, n) {
        const int l = (int)(i / D), d = (int)(i % D);
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += dlogits[(long long)b * L + l] * fmaxf(direct[((long long)b * L + l) * D + d], 0.f);
        dclf[i] = accumulate ? dclf[i] + s : s;
    }
}

// Both gradients of the head in one pass over `direct` (it is 115 MB at cfg2): a thread owns four columns d of one label
// and walks the notes in order, so dclf is summed in the same order as dclf_kernel does (bit-identical).
__global__ void logits_bwd_fused_kernel(const float* __restrict__ direct, const float* __restrict__ clf,
                                        const float* __restrict__ dlogits, float* __restrict__ ddirect,
                                        float* __restrict__ dclf, int B, int L, int D4, int accumulate) {
    const long long n = (long long)L * D4;
    const float4* dir4 = reinterpret_cast<const float4*>(direct);
    float4* dd4 = reinterpret_cast<float4*>(ddirect);
    GRID_STRIDE(i, n) {
        const int l = (int)(i / D4);
        const float4 c = reinterpret_cast<const float4*>(clf)[i];
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
        for (int b = 0; b < B; ++b) {
            const long long at = (long long)b * n + i;
            const float4 v = dir4[at];
            const float g = dlogits[(long long)b * L + l];
            float4 o;
            o.x = v.x > 0.f ? g * c.x : 0.f, o.y = v.y > 0.f ? g * c.y : 0.f;
            o.z = v.z > 0.f ? g * c.z : 0.f, o.w = v.w > 0.f ? g * c.w : 0.f;
            dd4[at] = o;
            s.x += g * fmaxf(v.x, 0.f), s.y += g * fmaxf(v.y, 0.f), s.z += g * fmaxf(v.z, 0.f), s.w += g * fmaxf(v.w, 0.f);
        }
        float4* out = reinterpret_cast<float4*>(dclf) + i;
        if (accumulate) {
            const float4 o = *out;
            s.x += o.x, s.y += o.y, s.z += o.z, s.w += o.w;
        }
        *out = s;
    }
}

// The head's backward in one pass over `direct`, one wave per label walking the notes in order (dclf is summed in the
// same order as dclf_kernel: bit-identical), plus the attention backward's row constant without reading dO or O:
//   delta[b,l] = sum_f dO[b,l,f] O[b,l,f],  dO = ddirect . Wo,  direct = O . Wo^T + bo
//              = sum_d ddirect[b,l,d] (direct[b,l,d] - bo[d])
// (icd_modules.py:142,159: the softmax-backward constant of the pooling that feeds output_proj).  NC = float4 chunks
// per lane: D <= 256 * NC.
template <int NC>
__global__ __launch_bounds__(256) void logits_bwd_delta_kernel(const float* __restrict__ direct,
                                                               const float* __restrict__ clf,
                                                               const float* __restrict__ bo,
                                                               const float* __restrict__ dlogits,
                                                               float* __restrict__ ddirect, float* __restrict__ dclf,
                                                               float* __restrict__ delta, int B, int L, int D4) {
    const int lane = threadIdx.x & 63;
    const int l = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (l >= L) return;
    float4 c[NC], bb[NC], s[NC];
#pragma unroll
    for (int u = 0; u < NC; ++u) {
        const int d4 = lane + 64 * u;
        const bool ok = d4 < D4;
        c[u] = ok ? reinterpret_cast<const float4*>(clf)[(long long)l * D4 + d4] : make_float4(0.f, 0.f, 0.f, 0.f);
        bb[u] = ok ? reinterpret_cast<const float4*>(bo)[d4] : make_float4(0.f, 0.f, 0.f, 0.f);
        s[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // four notes' rows are fetched before the first is consumed (the cross-lane sum at the end of a row would
    // otherwise keep the next row's loads from issuing)
    for (int b0 = 0; b0 < B; b0 += 4) {
        float4 v[4][NC];
        float g[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const long long row = (long long)min(b0 + q, B - 1) * L + l;
            g[q] = dlogits[row];
#pragma unroll
            for (int u = 0; u < NC; ++u) {
                const int d4 = min(lane + 64 * u, D4 - 1);
                v[q][u] = reinterpret_cast<const float4*>(direct)[row * D4 + d4];
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (b0 + q >= B) break;
            const long long row = (long long)(b0 + q) * L + l;
            float dl = 0.f;
#pragma unroll
            for (int u = 0; u < NC; ++u) {
                const int d4 = lane + 64 * u;
                if (d4 < D4) {
                    const float4 x = v[q][u];
                    float4 o;
                    o.x = x.x > 0.f ? g[q] * c[u].x : 0.f, o.y = x.y > 0.f ? g[q] * c[u].y : 0.f;
                    o.z = x.z > 0.f ? g[q] * c[u].z : 0.f, o.w = x.w > 0.f ? g[q] * c[u].w : 0.f;
                    reinterpret_cast<float4*>(ddirect)[row * D4 + d4] = o;
                    s[u].x += g[q] * fmaxf(x.x, 0.f), s[u].y += g[q] * fmaxf(x.y, 0.f);
                    s[u].z += g[q] * fmaxf(x.z, 0.f), s[u].w += g[q] * fmaxf(x.w, 0.f);
                    dl += o.x * (x.x - bb[u].x) + o.y * (x.y - bb[u].y) + o.z * (x.z - bb[u].z) + o.w * (x.w - bb[u].w);
                }
            }
            dl = wave_sum(dl);
            if (lane == 0) delta[row] = dl;
        }
    }
#pragma unroll
    for (int u = 0; u < NC; ++u) {
        const int d4 = lane + 64 * u;
        if (d4 < D4) reinterpret_cast<float4*>(dclf)[(long long)l * D4 + d4] = s[u];
    }
}

__global__ void ldam_bce_kernel(const float* __restrict__ logits, const float* __restrict__ y,
                                const float* __restrict__ margin, float C, float inv_b, float* loss,
                                float* __restrict__ dlogits, float* __restrict__ probs, int B, int L) {
    const long long n = (long long)B * L;
    float part = 0.f;
    GRID_STRIDE(i, n) {
        const int l = (int)(i % L);
        const float yy = y[i];
        const float lg = logits[i];
        float z = lg;
        if (margin) z -= yy * margin[l] * C;
        part += fmaxf(z, 0.f) - z * yy + log1pf(expf(-fabsf(z)));
        if (dlogits) dlogits[i] = (1.f / (1.f + expf(-z)) - yy) * inv_b;
        if (probs) probs[i] = 1.f / (1.f + expf(-lg));
    }
    part = wave_sum(part);
    __shared__ float wsum[EW_THREADS / 64];
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = part;
    __syncthreads();
    if (threadIdx.x == 0 && loss) {  // one atomic per block, waves added in a fixed order
        float t = 0.f;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += wsum[w];
        atomicAdd(loss, t * inv_b);
    }
}

__global__ void forward_final_kernel(const float* __restrict__ fc, const float* __restrict__ feats,
                                     const int64_t* __restrict__ labels, const float* __restrict__ margin, float C,
                                     float scale, float* loss, float* dfc, float* __restrict__ dfeat, int Bs, int D) {
    const int lane = threadIdx.x & 63;
    int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    for (; row < Bs; row += gridDim.x * (blockDim.x >> 6)) {
        const int64_t lab = labels[row];
        float s = 0.f;
        for (int d = lane; d < D; d += 64) s += fc[lab * D + d] * feats[(long long)row * D + d];
        s = wave_sum(s);
        float z = s;
        if (margin) z -= margin[lab] * C;
        const float dz = (1.f / (1.f + expf(-z)) - 1.f) * scale;
        if (lane == 0 && loss) atomicAdd(loss, (fmaxf(z, 0.f) - z + log1pf(expf(-fabsf(z)))) * scale);
        for (int d = lane; d < D; d += 64) {
            if (dfc) atomicAdd(dfc + lab * D + d, dz * feats[(long long)row * D + d]);
            if (dfeat) dfeat[(long long)row * D + d] = dz * fc[lab * D + d];
        }
    }
}

__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                             float* __restrict__ v, long long n, float lr, float b1, float b2, float eps, float wd) {
    GRID_STRIDE(i, n) {
        const float gg = g[i];
        const float mm = m[i] * b1 + (1.f - b1) * gg;
        const float vv = v[i] * b2 + (1.f - b2) * gg * gg;
        m[i] = mm;
        v[i] = vv;
        float upd = mm / (sqrtf(vv) + eps);
        if (wd > 0.f) upd += wd * p[i];
        p[i] -= lr * upd;
    }
}

// -------------------------------------------------------------------------------------------- helpers
__global__ void act_bwd_kernel(float* __restrict__ d, const float* __restrict__ y, long long n, int act) {
    GRID_STRIDE(i, n) {
        const float yy = y[i];
        float f = 1.f;
        if (act == GZSL_ACT_RELU) f = yy > 0.f ? 1.f : 0.f;
        else if (act == GZSL_ACT_TANH) f = 1.f - yy * yy;
        else if (act == GZSL_ACT_LRELU) f = yy > 0.f ? 1.f : 0.2f;
        d[i] *= f;
    }
}

// out[n] += sum_m x[m*ld + n]; block = 64 columns x 4 row-lanes, each block sweeps a row chunk
// V = columns per lane (4 with 16-byte loads when N, ld and the base allow it).  64 column lanes x 4 row lanes per block;
// eight independent row streams per thread keep the loads in flight.
template <int V>
__global__ void colsum_kernel(const float* __restrict__ x, float* out, float* partial, long long M, int N, int ld,
                              int rows_per_block) {
    typedef float vecV __attribute__((ext_vector_type(V)));
    __shared__ float part[4][64 * V];
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int col = (blockIdx.x * 64 + cx) * V;
    const long long r0 = (long long)blockIdx.y * rows_per_block;
    long long r1 = r0 + rows_per_block;
    if (r1 > M) r1 = M;
    float s[V];
#pragma unroll
    for (int v = 0; v < V; ++v) s[v] = 0.f;
    if (col < N) {
        long long r = r0 + ry;
        for (; r + 28 < r1; r += 32) {
            vecV t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = *reinterpret_cast<const vecV*>(x + (r + 4 * u) * ld + col);
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int v = 0; v < V; ++v) s[v] += t[u][v];
        }
        for (; r < r1; r += 4) {
            const vecV t = *reinterpret_cast<const vecV*>(x + r * ld + col);
#pragma unroll
            for (int v = 0; v < V; ++v) s[v] += t[v];
        }
    }
#pragma unroll
    for (int v = 0; v < V; ++v) part[ry][cx * V + v] = s[v];
    __syncthreads();
    if (ry == 0 && col < N) {
#pragma unroll
        for (int v = 0; v < V; ++v) {
            const int c = cx * V + v;
            const float t = part[0][c] + part[1][c] + part[2][c] + part[3][c];
            // same-address float atomics cost ~100 ns each (serialised at the memory side): with a workspace the
            // row blocks write partial sums and colsum_finish_kernel adds them up in a fixed order
            if (partial) partial[(long long)blockIdx.y * N + col + v] = t;
            else atomicAdd(out + col + v, t);
        }
    }
}

// 16 columns x 16 row lanes per block; fixed summation order
__global__ void colsum_finish_kernel(const float* __restrict__ partial, float* out, int nblk, int N) {
    __shared__ float part[16][17];
    const int cx = threadIdx.x & 15, ry = threadIdx.x >> 4;
    const int col = blockIdx.x * 16 + cx;
    float s = 0.f;
    if (col < N) {
        int b = ry;
        for (; b + 48 < nblk; b += 64) {
            float t[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) t[u] = partial[(long long)(b + 16 * u) * N + col];
#pragma unroll
            for (int u = 0; u < 4; ++u) s += t[u];
        }
        for (; b < nblk; b += 16) s += partial[(long long)b * N + col];
    }
    part[ry][cx] = s;
    __syncthreads();
    if (ry == 0 && col < N) {
        float t = 0.f;
#pragma unroll
        for (int u = 0; u < 16; ++u) t += part[u][cx];
        out[col] += t;
    }
}

// 32-bit index arithmetic (F*E*k < 2^31, checked by the launcher): the three 64-bit divisions per element made this
// 200k-element repack a 14 us kernel.
__global__ void conv_pack_kernel(const float* __restrict__ W, float* __restrict__ Wp, float* __restrict__ Wflip, int F,
                                 int E, int k) {
    const int n = F * E * k;
    const int ke = k * E;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int f = i / ke;
        const int rem = i - f * ke;
        const int e = rem / k;
        const int j = rem - e * k;
        const float w = W[i];
        if (Wp) Wp[f * ke + j * E + e] = w;
        if (Wflip) Wflip[e * (k * F) + (k - 1 - j) * F + f] = w;
    }
}

// -------------------------------------------------------------------------------------------- C ABI
#define ST ((hipStream_t)stream)

// Library-wide deterministic-reduction switch (include/gzsl_hip.h): reductions that normally end in a few float atomics
// (whose order is not fixed) take their fixed-order form instead.
static int g_deterministic = 0;
extern "C" int gzsl_set_deterministic(int on) {
    const int prev = g_deterministic;
    g_deterministic = on ? 1 : 0;
    return prev;
}

extern "C" int gzsl_emb_gather_fwd(const float* W, const int64_t* ids, const float* mask, float* out, long long rows,
                                   int E, long long row_offset, uint32_t seed, uint32_t drop_thresh, float drop_scale,
                                   void* stream) {
    if (rows <= 0) return 0;
    if ((E & 3) == 0 && (((uintptr_t)W | (uintptr_t)out) & 15) == 0) {
        hipLaunchKernelGGL(emb_gather_v4_kernel, dim3(ew_blocks(rows * (E / 4))), dim3(EW_THREADS), 0, ST, W, ids, mask, out,
                           rows, E / 4, row_offset, seed, drop_thresh, drop_scale);
        GZSL_CHECK_LAUNCH();
        return 0;
    }
    hipLaunchKernelGGL(emb_gather_kernel, dim3(ew_blocks(rows * E)), dim3(EW_THREADS), 0, ST, W, ids, mask, out, rows, E,
                       row_offset, seed, drop_thresh, drop_scale);
    GZSL_CHECK_LAUNCH();
    return 0;
}

extern "C" int gzsl_emb_scatter_add(float* dW, const int64_t* ids, const float* mask, const float* dOut, long long rows,
                                    int E, int64_t pad_idx, long long row_offset, uint32_t seed, uint32_t drop_thresh,
                                    float drop_scale, void* stream) {
    if (rows <= 0) return 0;
    hipLaunchKernelGGL(emb_scatter_kernel, dim3(ew_blocks(rows * E)), dim3(EW_THREADS), 0, ST, dW, ids, mask, dOut, rows,
                       E, pad_idx, row_offset, seed, drop_thresh, drop_scale);
    GZSL_CHECK_LAUNCH();
    return 0;
}

extern "C" int gzsl_emb_scatter_add_grouped(float* dW, const int64_t* ids, const float* mask, const float* dOut,
                                            long long rows, int E, int64_t pad_idx, long long row_offset,
                                            uint32_t seed, uint32_t drop_thresh, float drop_scale, void* stream) {
    if (rows <= 0) return 0;
    const size_t smem = sizeof(unsigned long long) * ES_CHUNK + sizeof(float) * ES_CHUNK * (size_t)E;
    if (E <= 0 || smem > 150 * 1024) return GZSL_EINVAL;  // E <= 297 (the reference uses 100 or 200)
    hipLaunchKernelGGL(emb_scatter_grouped_kernel, dim3((unsigned)((rows + ES_CHUNK - 1) / ES_CHUNK)), dim3(256), smem, ST,
                       dW, ids, mask, dOut, rows, E, pad_idx, row_offset, seed, drop_thresh, drop_scale);
    GZSL_CHECK_LAUNCH();
    return 0;
}

extern "C" int gzsl_label_avg_fwd(const float* W, const int64_t* idx, const float* mask, float* out, int N, int Td,
                                  int E, void* stream) {
    if (N <= 0) return 0;
    if ((E & 3) == 0 && (((uintptr_t)W | (uintptr_t)out) & 15) == 0) {
        hipLaunchKernelGGL(label_avg_fwd_v4_kernel, dim3(ew_blocks((long long)N * (E / 4))), dim3(EW_THREADS), 0, ST, W, idx,
                           mask, out, N, Td, E / 4);
        GZSL_CHECK_LAUNCH();
        return 0;
    }
    hipLaunchKernelGGL(label_avg_fwd_kernel, dim3(ew_blocks((long long)N * E)), dim3(EW_THREADS), 0, ST, W, idx, mask, out,
                       N, Td, E);
    GZSL_CHECK_LAUNCH();
    return 0;
}

extern "C" int gzsl_label_avg_bwd(float* dW, const float* davg, const int64_t* idx, const float* mask, int N, int Td,
                                  int E, int64_t pad_idx, void* stream) {
    if (N <= 0) return 0;
    hipLaunchKernelGGL(label_avg_bwd_kernel, dim3(ew_blocks((long long)N * E)), dim3(EW_THREADS), 0, ST, dW, davg, idx,
                       mask, N, Td, E, pad_idx);
    GZSL_CHECK_LAUNCH();
    return 0;
}

extern "C" int gzsl_spmm_csr(const int64_t* rowptr, const int64_t* col, const float* val, const float* x,
                             const float* pre, const float* post, float* out, int N, int H, int accumulate,
                             void* stream) {
    if (N <= 0) return 0;
    if (accumulate < 0 || accumulate > 2) return GZSL_EINVAL;
    // (atomic mode keeps one column per lane: a wave-instruction then adds 256 contiguous bytes, the shape float atomics
    // run at full rate in; four columns per lane would spread each instruction over 1 KiB in 4-byte pieces)
    if (accumulate != 2 && (H & 3) == 0 && (((uintptr_t)x | (uintptr_t)out) & 15) == 0) {
        hipLaunchKernelGGL(spmm_csr_v4_kernel, dim3(ew_blocks((long long)N * (H / 4))), dim3(EW_THREADS), 0, ST, rowptr, col,
                           val, x, pre, post, out, N, H / 4, accumulate);
        GZSL_CHECK_LAUNCH();
        return 0;
    }
    hipLaunchKernelGGL(spmm_csr_kernel, dim3(ew_blocks((long long)N * H)), dim3(EW_THREADS), 0, ST, rowptr, col, val, x,
                       pre, post, out, N, H, accumulate);
    GZSL_CHECK_LAUNCH();
    return 0;
}

extern "C" int gzsl_gru_gate_fwd(const float* gi, const float* gh, const float* h, float* hn, int N, int H,
                                 void* stream) {
    if (N <= 0) return 0;
    if ((H & 3) == 0 && (((uintptr_t)gi | (uintptr_t)gh | (uintptr_t)h | (uintptr_t)hn) & 15) == 0) {
        hipLaunchKernelGGL(gru_gate_fwd_v4_kernel, dim3(ew_blocks((long long)N * (H / 4))), dim3(EW_THREADS), 0, ST, gi, gh, h,
                           hn, N, H);
        GZSL_CHECK_LAUNCH();
        return 0;
    }
    hipLaunchKernelGGL(gru_gate_fwd_kernel, dim3(ew_blocks((long long)N * H)), dim3(EW_THREADS), 0, ST, gi, gh, h, hn, N,
                       H);
    GZSL_CHECK_LAUNCH();
    return 0;
}

extern "C" int gzsl_gru_seq_step(const float* gi, long long gi_stride, const float* gh, float* h, const float* mask, int T,
                                 int t, float* pool, int pool_mode, int Bg, int H, void* stream) {
    if (Bg <= 0) return 0;
    if (t < 0 || t >= T || pool_mode < 0 || pool_mode > 2 || (pool_mode != 0 && pool == nullptr)) return GZSL_EINVAL;
    hipLaunchKernelGGL(gru_seq_step_kernel, dim3(ew_blocks((long long)Bg * H)), dim3(EW_THREADS), 0, ST, gi, gi_stride, gh, h,
                       mask, T, t, pool, pool_mode, Bg, H);
    GZSL_CHECK_LAUNCH();
    return 0;
}

extern "C" int gzsl_gru_gate_bwd(const float* dhn, const float* gi, const float* gh, const float* h, float* dgi,
                                 float* dgh, float* dh, int N, int H, void* stream) {
    if (N <= 0) return 0;
    if ((H & 3) == 0 && (((uintptr_t)dhn | (uintptr_t)gi | (uintptr_t)gh | (uintptr_t)h | (uintptr_t)dgi | (uintptr_t)dgh |
                          (uintptr_t)dh) & 15) == 0) {
        hipLaunchKernelGGL(gru_gate_bwd_v4_kernel, dim3(ew_blocks((long long)N * (H / 4))), dim3(EW_THREADS), 0, ST, dhn, gi,
                           gh, h, dgi, dgh, dh, N, H);
        GZSL_CHECK_LAUNCH();
        return 0;
    }
    hipLaunchKernelGGL(gru_gate_bwd_kernel, dim3(ew_blocks((long long)N * H)), dim3(EW_THREADS), 0, ST, dhn, gi, gh, h,
                       dgi, dgh, dh, N, H);
    GZSL_CHECK_LAUNCH();
    return 0;
}

// ---- host batching -> device, eval scoring --------------------------------------------------------------
// utils/helper.py:117-123: the dense [B,L] multi-hot target built from each note's label list.  y is cleared by the
// launcher; lanes run along the ragged index list, duplicates just store 1 twice.
__global__ void multihot_kernel(const int64_t* __restrict__ idx, int W, float* __restrict__ y, int B, int L) {
    const long long n = (long long)B * W;
    GRID_STRIDE(i, n) {
        const long long b = i / W;
        const int64_t c = idx[i];
        if (c >= 0 && c < L) y[b * L + c] = 1.f;
    }
}

// utils/metrics.py:29,63-121 reduced to what the macro/micro accuracy, precision, recall and F1 need: per label column,
// the number of notes with yhat&y, yhat, y, yhat|y, where yhat = round(sigmoid(logit)) (half-to-even: exactly 0.5 -> 0).
// One thread per label, rows walked in order (a wave reads 256 consecutive bytes of a row): no atomics, exact integers.
__global__ void eval_counts_kernel(const float* __restrict__ logits, int ldz, const float* __restrict__ y, int ldy,
                                   int* __restrict__ counts, float* __restrict__ probs, int ldp, int B, int Lc) {
    GRID_STRIDE(l, Lc) {
        int tp = 0, pp = 0, pos = 0, un = 0;
        for (int b = 0; b < B; ++b) {
            const float p = 1.f / (1.f + expf(-logits[(long long)b * ldz + l]));
            const int h = p > 0.5f, t = y[(long long)b * ldy + l] != 0.f;
            tp += h & t, pp += h, pos += t, un += h | t;
            if (probs) probs[(long long)b * ldp + l] = p;
        }
        counts[l] += tp, counts[Lc + l] += pp, counts[2ll * Lc + l] += pos, counts[3ll * Lc + l] += un;
    }
}

extern "C" int gzsl_logits_fwd(const float* direct, const float* clf, float* logits, int B, int L, int D,
                               void* stream) {
    if (B <= 0 || L <= 0) return 0;
    if ((D & 3) == 0 && (((uintptr_t)direct | (uintptr_t)clf) & 15) == 0) {
        hipLaunchKernelGGL(logits_fwd_v4_kernel, dim3(ew_blocks((long long)B * L, 4)), dim3(EW_THREADS), 0, ST, direct, clf,
                           logits, B, L, D / 4);
        GZSL_CHECK_LAUNCH();
        return 0;
    }
    hipLaunchKernelGGL(logits_fwd_kernel, dim3(ew_blocks((long long)B * L, 4)), dim3(EW_THREADS), 0, ST, direct, clf,
                       logits, B, L, D);
    GZSL_CHECK_LAUNCH();
    return 0;
}

extern "C" int gzsl_logits_bwd(const float* direct, const float* clf, const float* dlogits, float* ddirect,
                               float* dclf, int B, int L, int D, int accumulate_dclf, void* stream) {
    if (B <= 0 || L <= 0) return 0;
    if (ddirect && dclf && (D & 3) == 0 &&
        (((uintptr_t)direct | (uintptr_t)clf | (uintptr_t)ddirect | (uintptr_t)dclf) & 15) == 0) {
        hipLaunchKernelGGL(logits_bwd_fused_kernel, dim3(ew_blocks((long long)L * (D / 4))), dim3(EW_THREADS), 0, ST, direct,
                           clf, dlogits, ddirect, dclf, B, L, D / 4, accumulate_dclf);
        GZSL_CHECK_LAUNCH();
        return 0;
    }
    if (ddirect) {
        hipLaunchKernelGGL(logits_bwd_kernel, dim3(ew_blocks((long long)B * L * D)), dim3(EW_THREADS), 0, ST, direct, clf,
                           dlogits, ddirect, B, L, D);
        GZSL_CHECK_LAUNCH();
    }
    if (dclf) {
        hipLaunchKernelGGL(dclf_kernel, dim3(ew_blocks((long long)L * D)), dim3(EW_THREADS), 0, ST, direct, dlogits, dclf,
                           B, L, D, accumulate_dclf);
        GZSL_CHECK_LAUNCH();
    }
    return 0;
}

extern "C" int gzsl_logits_bwd_delta(const float* direct, const float* clf, const float* bo, const float* dlogits,
                                     float* ddirect, float* dclf, float* delta, int B, int L, int D, void* stream) {
    if (B <= 0 || L <= 0) return 0;
    if ((D & 3) != 0 || D > 1024 ||
        (((uintptr_t)direct | (uintptr_t)clf | (uintptr_t)bo | (uintptr_t)ddirect | (uintptr_t)dclf) & 15) != 0)
        return GZSL_EINVAL;
    const dim3 grid(cdiv(L, 4)), block(256);
    const int D4 = D / 4;
#define GZ_LBD(NC) \
    hipLaunchKernelGGL(logits_bwd_delta_kernel<NC>, grid, block, 0, ST, direct, clf, bo, dlogits, ddirect, dclf, delta, B, L, D4)
    if (D4 <= 64) GZ_LBD(1);
    else if (D4 <= 128) GZ_LBD(2);
    else GZ_LBD(4);
#undef GZ_LBD
    GZSL_CHECK_LAUNCH();
    return 0;
}

extern "C" int gzsl_ldam_bce(const float* logits, const float* y, const float* margin, float C, float inv_b,
                             float* loss, float* dlogits, float* probs, int B, int L, void* stream) {
    if (B <= 0 || L <= 0) return 0;
    // every block ends in ONE float atomic on *loss, and same-address atomics serialise at the memory side (~30 ns each):
    // 128 blocks are plenty for the <= 1.7 MB this kernel moves (563 blocks cost 20 us at cfg2, 128 cost 6)
    int blocks = ew_blocks((long long)B * L);
    if (blocks > 128) blocks = 128;
    if (g_deterministic) blocks = 1;  // one block: its waves' sums are added in a fixed order, one add into *loss
    hipLaunchKernelGGL(ldam_bce_kernel, dim3(blocks), dim3(EW_THREADS), 0, ST, logits, y, margin, C, inv_b, loss, dlogits,
                       probs, B, L);
    GZSL_CHECK_LAUNCH();
    return 0;
}

extern "C" int gzsl_forward_final(const float* fc, const float* feats, const int64_t* labels, const float* margin,
                                  float C, float scale, float* loss, float* dfc, float* dfeat, int Bs, int D,
                                  void* stream) {
    if (Bs <= 0) return 0;
    hipLaunchKernelGGL(forward_final_kernel, dim3(ew_blocks(Bs, 4)), dim3(EW_THREADS), 0, ST, fc, feats, labels, margin, C,
                       scale, loss, dfc, dfeat, Bs, D);
    GZSL_CHECK_LAUNCH();
    return 0;
}

extern "C" int gzsl_adamw_nobias_step(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1,
                                      float beta2, float eps, float wd, void* stream) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(adamw_kernel, dim3(ew_blocks(n)), dim3(EW_THREADS), 0, ST, p, g, m, v, n, lr, beta1, beta2, eps, wd);
    GZSL_CHECK_LAUNCH();
    return 0;
}

extern "C" int gzsl_act_bwd(float* d, const float* y, long long n, int act, void* stream) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(act_bwd_kernel, dim3(ew_blocks(n)), dim3(EW_THREADS), 0, ST, d, y, n, act);
    GZSL_CHECK_LAUNCH();
    return 0;
}

static int colsum_launch(const float* x, float* out, long long M, int N, int ld, float* work, long long work_floats,
                         hipStream_t st) {
    if (M <= 0 || N <= 0) return 0;
    int rows_per_block = 128;
    long long gy = (M + rows_per_block - 1) / rows_per_block;
    long long cap = 4096;
    if (work && work_floats / N < cap) cap = work_floats / N;
    if (cap < 1) work = nullptr, cap = 4096;
    if (gy > cap) {
        rows_per_block = (int)((M + cap - 1) / cap);
        gy = (M + rows_per_block - 1) / rows_per_block;
    }
    if (gy <= 8 && !g_deterministic) work = nullptr;  // a handful of atomics per column is cheaper than a second launch
    if ((N & 3) == 0 && (ld & 3) == 0 && ((uintptr_t)x & 15) == 0)
        hipLaunchKernelGGL(colsum_kernel<4>, dim3(cdiv(N, 256), (unsigned)gy), dim3(256), 0, st, x, out, work, M, N, ld,
                           rows_per_block);
    else
        hipLaunchKernelGGL(colsum_kernel<1>, dim3(cdiv(N, 64), (unsigned)gy), dim3(256), 0, st, x, out, work, M, N, ld,
                           rows_per_block);
    if (work) hipLaunchKernelGGL(colsum_finish_kernel, dim3(cdiv(N, 16)), dim3(256), 0, st, work, out, (int)gy, N);
    GZSL_CHECK_LAUNCH();
    return 0;
}

extern "C" int gzsl_colsum(const float* x, float* out, long long M, int N, int ld, void* stream) {
    return colsum_launch(x, out, M, N, ld, nullptr, 0, ST);
}

extern "C" int gzsl_colsum_ws(const float* x, float* out, long long M, int N, int ld, float* work, long long work_floats,
                              void* stream) {
    return colsum_launch(x, out, M, N, ld, work, work_floats, ST);
}

// ---- allocation-free column sum (no counterpart in the reference; see include/gzsl_hip.h) ---------------------------
// The partial sums of gzsl_colsum_ws live in a file-scope device array instead of a caller workspace.  Each GPU the
// library runs on holds its own copy at its own address until the process ends; in host code the array's NAME is only a
// stand-in for hipGetSymbolAddress (handing it to a kernel would fault), so the address is looked up once per device
// and kept.
#define GZ_SCRATCH_FLOATS (GZSL_COLSUM_STATIC_MAX_N * 64)  // 2 MiB: >= 64 row blocks of partial sums at the widest N
static __device__ __align__(256) float gz_colsum_scratch[GZ_SCRATCH_FLOATS];
#define GZ_MAX_DEVICES 64
static float* g_scratch_ptr[GZ_MAX_DEVICES];  // per device; written once under g_scratch_mu
static std::mutex g_scratch_mu;

// The scratch array is looked up for the CURRENT device; the kernels run on `st`'s device.  A stream of another device
// would hand the kernels a foreign GPU's scratch address: refuse (the null stream always belongs to the current device).
static int scratch_for_current_device(float** out, hipStream_t st) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    if (dev < 0 || dev >= GZ_MAX_DEVICES) return GZSL_EINVAL;
    if (st) {
        int sdev = -1;
        e = hipStreamGetDevice(st, &sdev);
        if (e != hipSuccess) return (int)e;
        if (sdev != dev) return GZSL_EINVAL;
    }
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    if (!g_scratch_ptr[dev]) {
        float* p = nullptr;
        e = hipGetSymbolAddress((void**)&p, HIP_SYMBOL(gz_colsum_scratch));
        if (e != hipSuccess) return (int)e;
        if (!p) return GZSL_EINVAL;
        g_scratch_ptr[dev] = p;
    }
    *out = g_scratch_ptr[dev];
    return 0;
}

extern "C" int gzsl_colsum_static(const float* x, float* out, long long M, int N, int ld, void* stream) {
    if (M <= 0 || N <= 0) return 0;
    if (N > GZSL_COLSUM_STATIC_MAX_N) return GZSL_ETOOBIG;
    float* work = nullptr;
    const int rc = scratch_for_current_device(&work, ST);
    if (rc != 0) return rc;
    // every partial-sum word the finish kernel reads is written by this call's first kernel: nothing to clear
    return colsum_launch(x, out, M, N, ld, work, GZ_SCRATCH_FLOATS, ST);
}

extern "C" int gzsl_multihot(const int64_t* y_indices, int W, float* y, int B, int L, void* stream) {
    if (B <= 0 || L <= 0) return 0;
    hipError_t e = hipMemsetAsync(y, 0, (size_t)B * L * sizeof(float), ST);
    if (e != hipSuccess) return (int)e;
    if (W <= 0) return 0;
    hipLaunchKernelGGL(multihot_kernel, dim3(ew_blocks((long long)B * W)), dim3(EW_THREADS), 0, ST, y_indices, W, y, B, L);
    GZSL_CHECK_LAUNCH();
    return 0;
}

extern "C" int gzsl_eval_counts(const float* logits, int ld_logits, const float* y, int ld_y, int* counts, float* probs,
                                int ld_probs, int B, int Lc, void* stream) {
    if (B <= 0 || Lc <= 0) return 0;
    hipLaunchKernelGGL(eval_counts_kernel, dim3(ew_blocks(Lc, 64)), dim3(64), 0, ST, logits, ld_logits, y, ld_y, counts,
                       probs, ld_probs, B, Lc);
    GZSL_CHECK_LAUNCH();
    return 0;
}

extern "C" int gzsl_conv_weight_pack(const float* W, float* Wp, float* Wflip, int F, int E, int k, void* stream) {
    if ((long long)F * E * k >= (1ll << 31)) return GZSL_EINVAL;
    hipLaunchKernelGGL(conv_pack_kernel, dim3(ew_blocks((long long)F * E * k)), dim3(EW_THREADS), 0, ST, W, Wp, Wflip, F,
                       E, k);
    GZSL_CHECK_LAUNCH();
    return 0;
}

extern "C" const char* gzsl_build_info(void) { return "libgzsl_hip gfx950 (MI355X) fp32-mfma16x16x4"; }
