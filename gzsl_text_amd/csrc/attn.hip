// Fused per-label attention, forward and backward (gfx950, exact-fp32 MFMA 16x16x4).
//
// Replaces reference icd_modules.py:137-142 (conv_attn):
//     S[b,l,t] = <proj_x[b,t,:], q[l,:]>;  P = softmax_t(S);  Pd = dropout_{0.2}(P);  O[b,l,:] = sum_t Pd[b,l,t] x[b,t,:]
// which materialises B x L x T' (1.4 GB at B16/L9000/T2491) several times.  Here the scores never leave
// registers: forward keeps a running (max, sum) per label (online softmax over token tiles), backward recomputes
// P from the saved log-sum-exp.  No padding mask is applied (the reference applies none: padded positions carry
// relu(conv bias)-driven keys and DO receive attention mass).
//
// Data layout (all row-major fp32, as the reference's tensors):
//   PX  [B, Tp, E]  keys   = tanh(proj(x))          X [B, Tp, F]  values = relu(conv(emb))
//   Q   [L, E]      queries = average label embedding rows of this rank's label shard
//   O   [B, L, F],  LSE [B, L]
//
// MFMA orientation (v_mfma_f32_16x16x4_f32: lane l holds A[l&15][l>>4], B[l>>4][l&15]; D reg r = row 4*(l>>4)+r,
// col l&15).  Forward puts LABELS on the lane/column axis:  S^T tile = K_tile . Q^T  (rows = tokens), so the
// exponentiated tile is already the B operand of  O^T += V^T . P^T  -- no LDS round trip, no transpose.
// Backward puts TOKENS on the lane axis (a workgroup owns 128 tokens and sweeps the labels), so S and dP tiles are
// directly the B operands of dV^T += dO^T . Pd and dK^T += Q^T . dS; only dS crosses LDS once (per wave) for dQ.
#include "common.h"
#include "rng.h"
#include "../../include/gzsl_hip.h"

#define FWD_BT 32          // tokens per LDS tile (two 16-token MFMA sub-tiles)
#define FWD_WAVES 4        // 64 labels per workgroup
#define RESCALE_THR 12.0f  // lazy online-softmax rescale: fp32 P may grow to e^12 before the accumulators are rescaled

struct AttnFwdArgs {
    const float* PX;
    const float* X;
    const float* Q;
    float* O;
    float* LSE;
    int B, Tp, E, F, L;
    int l_offset, L_total;  // global label coordinates for the dropout hash
    uint32_t seed, thresh;
    float scale;
};

template <int ES, int NFT>
__global__ __launch_bounds__(FWD_WAVES * 64) void label_attn_fwd_kernel(AttnFwdArgs a) {
    constexpr int EP = 4 * ES + 2;    // K row stride: 2*(odd) -> 16 rows x 2 k-lanes hit 32 distinct banks
    constexpr int FP = 16 * NFT + 4;  // V row stride: = 4 mod 8 -> rows 4q and 4q+4 land 16 banks apart
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* Ks = reinterpret_cast<float*>(smem_raw);
    float* Vs = Ks + FWD_BT * EP;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, lq = lane >> 4;
    const int nLT = (a.L + 16 * FWD_WAVES - 1) / (16 * FWD_WAVES);
    // XCD-aware mapping: blocks with equal blockIdx % 8 share an XCD (and its 4 MiB L2); give them the same note b
    // so that note's keys/values (Tp*(E+F)*4 B ~ 3 MB) stay L2-resident while its label tiles sweep them.
    int b, lt;
    {
        const int id = blockIdx.x;
        if ((a.B & 7) == 0) {
            const int xcd = id & 7, slot = id >> 3;
            b = (slot / nLT) * 8 + xcd;
            lt = slot % nLT;
        } else {
            b = id / nLT;
            lt = id % nLT;
        }
    }
    const int lbl = lt * 16 * FWD_WAVES + wave * 16 + lr;  // this lane's label (column)
    const bool lbl_ok = lbl < a.L;

    // Q^T fragment: B operand of S^T = K.Q^T ; qreg[s] = Q[lbl][4s + lq]
    float qreg[ES];
#pragma unroll
    for (int s = 0; s < ES; ++s) {
        int e = 4 * s + lq;
        qreg[s] = (lbl_ok && e < a.E) ? a.Q[(long long)lbl * a.E + e] : 0.f;
    }
    // zero the K pad columns once (tile loads never touch them) so 0 * garbage cannot make NaN
    for (int i = tid; i < FWD_BT * (EP - a.E); i += FWD_WAVES * 64) {
        int r = i / (EP - a.E), c = a.E + i % (EP - a.E);
        Ks[r * EP + c] = 0.f;
    }

    const uint32_t rowkey = gz_row_key(a.seed, GZSL_STREAM_ATTN, (uint32_t)b * (uint32_t)a.L_total + (uint32_t)(a.l_offset + lbl));
    f32x4 acc[NFT];
#pragma unroll
    for (int j = 0; j < NFT; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m = -INFINITY, lsum = 0.f;

    const float* PXb = a.PX + (long long)b * a.Tp * a.E;
    const float* Xb = a.X + (long long)b * a.Tp * a.F;

    for (int t0 = 0; t0 < a.Tp; t0 += FWD_BT) {
        __syncthreads();  // previous tile fully consumed
        // stage K/V rows t0..t0+31 (rows past Tp are zero-filled: 0 * P must stay 0)
        for (int r = wave; r < FWD_BT; r += FWD_WAVES) {
            const int t = t0 + r;
            const bool ok = t < a.Tp;
            for (int c = lane; c < a.E; c += 64) Ks[r * EP + c] = ok ? PXb[(long long)t * a.E + c] : 0.f;
            for (int c = lane; c < 16 * NFT; c += 64) Vs[r * FP + c] = (ok && c < a.F) ? Xb[(long long)t * a.F + c] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < FWD_BT / 16; ++u) {
            // S^T sub-tile: rows = tokens t0+16u+4lq+r, cols = labels
            f32x4 s0 = f32x4{0.f, 0.f, 0.f, 0.f}, s1 = s0;
            const float* kp = Ks + (16 * u + lr) * EP + lq;
#pragma unroll
            for (int s = 0; s < ES; ++s) {
                if (s & 1) s1 = mfma16(kp[4 * s], qreg[s], s1);
                else s0 = mfma16(kp[4 * s], qreg[s], s0);
            }
            float sv[4];
            float mt = -INFINITY;
            const int tbase = t0 + 16 * u + 4 * lq;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = s0[r] + s1[r];
                sv[r] = (tbase + r < a.Tp) ? v : -INFINITY;
                mt = fmaxf(mt, sv[r]);
            }
            mt = fmaxf(mt, __shfl_xor(mt, 16));
            mt = fmaxf(mt, __shfl_xor(mt, 32));
            if (__any(mt > m + RESCALE_THR)) {  // wave-uniform; always taken on the first tile (m = -inf)
                const float mn = fmaxf(m, mt);
                const float alpha = __expf(m - mn);  // m = -inf -> 0
#pragma unroll
                for (int j = 0; j < NFT; ++j) acc[j] *= alpha;
                lsum *= alpha;
                m = mn;
            }
            float p[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                p[r] = __expf(sv[r] - m);
                lsum += p[r];
                if (a.thresh) {
                    const uint32_t uu = gz_u32(rowkey, (uint32_t)(tbase + r));
                    p[r] = (uu >= a.thresh) ? p[r] * a.scale : 0.f;
                }
            }
            // O^T[f, label] += sum_t V[t, f] * P[t, label]; k-step s' pairs tokens {4q + s'}
            const float* vp = Vs + (16 * u + 4 * lq) * FP + lr;
#pragma unroll
            for (int j = 0; j < NFT; ++j) {
#pragma unroll
                for (int sp = 0; sp < 4; ++sp) acc[j] = mfma16(vp[sp * FP + 16 * j], p[sp], acc[j]);
            }
        }
    }
    lsum += __shfl_xor(lsum, 16);
    lsum += __shfl_xor(lsum, 32);
    if (lbl_ok) {
        const float inv = 1.0f / lsum;
        if (lq == 0) a.LSE[(long long)b * a.L + lbl] = m + __logf(lsum);
        float* Op = a.O + ((long long)b * a.L + lbl) * a.F;
        const bool vec = (a.F & 3) == 0;
#pragma unroll
        for (int j = 0; j < NFT; ++j) {
            const int f = 16 * j + 4 * lq;
            if (vec && f + 3 < a.F) {
                *reinterpret_cast<float4*>(Op + f) =
                    make_float4(acc[j][0] * inv, acc[j][1] * inv, acc[j][2] * inv, acc[j][3] * inv);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (f + r < a.F) Op[f + r] = acc[j][r] * inv;
            }
        }
    }
}

// delta[b,l] = sum_f dO[b,l,f] * O[b,l,f]   (= sum_t P dP, the softmax-backward row constant)
__global__ void attn_delta_kernel(const float* __restrict__ dO, const float* __restrict__ O, float* __restrict__ delta,
                                  long long rows, int F) {
    const int lane = threadIdx.x & 63;
    long long row = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long long stride = (long long)gridDim.x * (blockDim.x >> 6);
    for (; row < rows; row += stride) {
        float s = 0.f;
        for (int f = lane; f < F; f += 64) s += dO[row * F + f] * O[row * F + f];
        s = wave_sum(s);
        if (lane == 0) delta[row] = s;
    }
}

#define BWD_STW (16 * 7 + 1)  // staging row: up to 7 tiles (112 columns) per pass, odd stride

// Accumulator tiles hold [column-of-output (4*lq + r), token (lr)]; write them to the wave's private LDS staging as
// [token][column] and add whole token rows to global memory (256 contiguous bytes per wave-instruction: the shape
// float atomics run at full rate in).  All indices are compile-time after unrolling (no scratch).
template <int NTILES>
__device__ __forceinline__ void flush_tiles(const f32x4 (&tiles)[NTILES], float* st, int lane, int tok0, int Tp,
                                            int width, float* out, int ld) {
    const int lr = lane & 15, lq = lane >> 4;
#pragma unroll
    for (int j0 = 0; j0 < NTILES; j0 += 7) {
        constexpr int dummy = 0;
        (void)dummy;
        const int nj = (NTILES - j0) < 7 ? (NTILES - j0) : 7;
#pragma unroll
        for (int jj = 0; jj < 7; ++jj) {
            if (jj < nj) {
#pragma unroll
                for (int r = 0; r < 4; ++r) st[lr * BWD_STW + 16 * jj + 4 * lq + r] = tiles[j0 + jj < NTILES ? j0 + jj : 0][r];
            }
        }
        for (int trow = 0; trow < 16; ++trow) {
            const int t = tok0 + trow;
            if (t >= Tp) break;
            for (int c = lane; c < 16 * nj; c += 64) {
                const int col = 16 * j0 + c;
                if (col < width) atomicAdd(out + (long long)t * ld + col, st[trow * BWD_STW + c]);
            }
        }
    }
}

#define BWD_BL 16  // labels per pipeline step

struct AttnBwdArgs {
    const float* dO;
    const float* PX;
    const float* X;
    const float* Q;
    const float* LSE;
    const float* DELTA;
    float* dPX;  // [B,Tp,E]  accumulated with float atomics (zero-initialised by the caller)
    float* dX;   // [B,Tp,F]  accumulated with float atomics
    float* dQ;   // [L,E]     accumulated with float atomics
    int B, Tp, E, F, L;
    int l_offset, L_total;
    int nsplit;  // label splits per (b, token tile): raises the workgroup count past the CU count
    uint32_t seed, thresh;
    float scale;
};

// LDS plan of the backward kernel, shared by the kernel and its launcher
template <int ES, int NFT, int NW>
struct BwdLds {
    static constexpr int NET = (4 * ES + 15) / 16;                        // 16-wide tiles over E
    static constexpr int EPQ = (16 * NET + 2 > 4 * ES + 2) ? 16 * NET + 2 : 4 * ES + 2;  // Q / K row stride (2*odd)
    static constexpr int FP = 16 * NFT + 2;                               // dO row stride (2*odd)
    static constexpr int SCR = 17;
    static constexpr int DQW = 16 * NET;
    static constexpr int oQ = 0;                                          // [2][16][EPQ]
    static constexpr int odO = oQ + 2 * BWD_BL * EPQ;                     // [2][16][FP]
    static constexpr int oLse = odO + 2 * BWD_BL * FP;                    // [2][16]
    static constexpr int oDel = oLse + 2 * BWD_BL;                        // [2][16]
    static constexpr int oScr = oDel + 2 * BWD_BL;                        // [NW][16][17]
    static constexpr int odQ = oScr + NW * 16 * SCR;                      // [2][16][DQW]
    static constexpr int oK = odQ + 2 * BWD_BL * DQW;                     // [NW*16][EPQ]  this workgroup's keys
    static constexpr int total = oK + NW * 16 * EPQ;
    static constexpr int stage = NW * 16 * BWD_STW;                       // epilogue staging (aliases the front)
    static constexpr int floats = total > stage ? total : stage;
};

template <int ES, int NFT, int NW>
__global__ __launch_bounds__(NW * 64) void label_attn_bwd_kernel(AttnBwdArgs a) {
    using P = BwdLds<ES, NFT, NW>;
    constexpr int NET = P::NET, FS = 4 * NFT, EPQ = P::EPQ, FP = P::FP, SCR = P::SCR, DQW = P::DQW;
    constexpr int NT = NW * 64;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* sm = reinterpret_cast<float*>(smem_raw);
    float* Qs = sm + P::oQ;
    float* dOs = sm + P::odO;
    float* lse_s = sm + P::oLse;
    float* del_s = sm + P::oDel;
    float* dQs = sm + P::odQ;
    float* Kt = sm + P::oK;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, lq = lane >> 4;
    const int nTT = (a.Tp + 16 * NW - 1) / (16 * NW);
    int b, tt, sp;
    {
        int id = blockIdx.x;
        sp = id % a.nsplit;
        id /= a.nsplit;
        if ((a.B & 7) == 0) {  // same-XCD blocks share a note b (its dO rows and keys/values stay in that L2)
            const int xcd = id & 7, slot = id >> 3;
            b = (slot / nTT) * 8 + xcd;
            tt = slot % nTT;
        } else {
            b = id / nTT;
            tt = id % nTT;
        }
    }
    const int tok0 = tt * 16 * NW + wave * 16;
    const int tok = tok0 + lr;  // this lane's token (column of S / dP)
    const bool tok_ok = tok < a.Tp;
    // label range of this split, in units of 16-label steps
    const int nsteps_all = (a.L + BWD_BL - 1) / BWD_BL;
    const int steps_per = (nsteps_all + a.nsplit - 1) / a.nsplit;
    const int step_beg = sp * steps_per;
    const int step_end = min(nsteps_all, step_beg + steps_per);

    const float* PXb = a.PX + (long long)b * a.Tp * a.E;
    const float* Xb = a.X + (long long)b * a.Tp * a.F;

    // zero all of LDS once: pad columns of the staged tiles must never hold NaN bit patterns
    for (int i = tid; i < P::floats; i += NT) sm[i] = 0.f;
    __syncthreads();
    // this workgroup's keys: Kt[token][e] (B operand of S = Q.K^T and of dQ += dS.K)
    for (int i = tid; i < NW * 16 * 4 * ES; i += NT) {
        const int r = i / (4 * ES), c = i % (4 * ES);
        const int t = tt * 16 * NW + r;
        if (t < a.Tp && c < a.E) Kt[r * EPQ + c] = PXb[(long long)t * a.E + c];
    }
    // values stay in registers: B operand of dP = dO.V^T :  V[tok][4s + lq]
    float vT[FS];
#pragma unroll
    for (int s = 0; s < FS; ++s) {
        int f = 4 * s + lq;
        vT[s] = (tok_ok && f < a.F) ? Xb[(long long)tok * a.F + f] : 0.f;
    }

    f32x4 dKT[NET], dVT[NFT];
#pragma unroll
    for (int j = 0; j < NET; ++j) dKT[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NFT; ++j) dVT[j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const uint32_t seedmix = gz_mix32(a.seed + GZSL_STREAM_ATTN * 0x632BE5ABu);
    const uint32_t rowbase = (uint32_t)b * (uint32_t)a.L_total + (uint32_t)a.l_offset;

    // staging: thread -> (row, col) pieces of the 16 x E and 16 x F tiles
    constexpr int QPT = (BWD_BL * 4 * ES + NT - 1) / NT;
    constexpr int OPT = (BWD_BL * 16 * NFT + NT - 1) / NT;
    float pq[QPT], po[OPT], plse = 0.f, pdel = 0.f;
    auto prefetch = [&](int step) {
        const int l0 = step * BWD_BL;
#pragma unroll
        for (int i = 0; i < QPT; ++i) {
            int idx = tid + NT * i, r = idx / (4 * ES), c = idx % (4 * ES);
            int l = l0 + r;
            pq[i] = (r < BWD_BL && l < a.L && c < a.E) ? a.Q[(long long)l * a.E + c] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < OPT; ++i) {
            int idx = tid + NT * i, r = idx / (16 * NFT), c = idx % (16 * NFT);
            int l = l0 + r;
            po[i] = (r < BWD_BL && l < a.L && c < a.F) ? a.dO[((long long)b * a.L + l) * a.F + c] : 0.f;
        }
        if (tid < BWD_BL) {
            int l = l0 + tid;
            plse = (l < a.L) ? a.LSE[(long long)b * a.L + l] : INFINITY;  // +inf -> P = 0 for labels past L
            pdel = (l < a.L) ? a.DELTA[(long long)b * a.L + l] : 0.f;
        }
    };
    auto commit = [&](int buf) {
#pragma unroll
        for (int i = 0; i < QPT; ++i) {
            int idx = tid + NT * i, r = idx / (4 * ES), c = idx % (4 * ES);
            if (r < BWD_BL) Qs[(buf * BWD_BL + r) * EPQ + c] = pq[i];
        }
#pragma unroll
        for (int i = 0; i < OPT; ++i) {
            int idx = tid + NT * i, r = idx / (16 * NFT), c = idx % (16 * NFT);
            if (r < BWD_BL) dOs[(buf * BWD_BL + r) * FP + c] = po[i];
        }
        if (tid < BWD_BL) {
            lse_s[buf * BWD_BL + tid] = plse;
            del_s[buf * BWD_BL + tid] = pdel;
        }
    };

    if (step_beg < step_end) {
        prefetch(step_beg);
        commit(0);
    }
    __syncthreads();

    float* myscr = sm + P::oScr + wave * 16 * SCR;
    const float* Kw = Kt + (wave * 16) * EPQ;  // this wave's 16 keys
    for (int step = step_beg; step < step_end; ++step) {
        const int buf = (step - step_beg) & 1;
        const int l0 = step * BWD_BL;
        const bool more = step + 1 < step_end;
        if (more) prefetch(step + 1);
        // flush the dQ partials of the previous step (other buffer) with row-contiguous float atomics; every thread
        // clears exactly the words it read (pad columns only ever receive zeros), so no extra barrier is needed
        if (step > step_beg) {
            float* src = dQs + (buf ^ 1) * BWD_BL * DQW;
            const int lprev = l0 - BWD_BL;
            for (int i = tid; i < BWD_BL * a.E; i += NT) {
                int r = i / a.E, c = i % a.E;
                const float v = src[r * DQW + c];
                src[r * DQW + c] = 0.f;
                if (lprev + r < a.L) atomicAdd(a.dQ + (long long)(lprev + r) * a.E + c, v);
            }
        }
        const float* Qt = Qs + buf * BWD_BL * EPQ;
        const float* dOt = dOs + buf * BWD_BL * FP;
        // S tile (rows = labels l0 + 4lq + r, cols = tokens) and dP tile, two accumulators each (40-cycle MFMA latency)
        f32x4 s0 = f32x4{0.f, 0.f, 0.f, 0.f}, s1 = s0, d0 = s0, d1 = s0;
        {
            const float* qp = Qt + lr * EPQ + lq;
            const float* kp = Kw + lr * EPQ + lq;
#pragma unroll
            for (int s = 0; s < ES; ++s) {
                if (s & 1) s1 = mfma16(qp[4 * s], kp[4 * s], s1);
                else s0 = mfma16(qp[4 * s], kp[4 * s], s0);
            }
            const float* op = dOt + lr * FP + lq;
#pragma unroll
            for (int s = 0; s < FS; ++s) {
                if (s & 1) d1 = mfma16(op[4 * s], vT[s], d1);
                else d0 = mfma16(op[4 * s], vT[s], d0);
            }
        }
        float pd[4], ds[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int lrow = 4 * lq + r;
            const float lse = lse_s[buf * BWD_BL + lrow];
            const float del = del_s[buf * BWD_BL + lrow];
            float p = tok_ok ? __expf(s0[r] + s1[r] - lse) : 0.f;
            float dsc = 1.f;
            if (a.thresh) {
                const uint32_t key = gz_mix32(seedmix ^ ((rowbase + (uint32_t)(l0 + lrow)) * 0x9E3779B1u));
                dsc = (gz_u32(key, (uint32_t)tok) >= a.thresh) ? a.scale : 0.f;
            }
            pd[r] = p * dsc;
            ds[r] = p * ((d0[r] + d1[r]) * dsc - del);
        }
        // dV^T[f, tok] += sum_l dO[l, f] Pd[l, tok] ;  dK^T[e, tok] += sum_l Q[l, e] dS[l, tok]
        {
            const float* op = dOt + (4 * lq) * FP + lr;
#pragma unroll
            for (int j = 0; j < NFT; ++j)
#pragma unroll
                for (int s = 0; s < 4; ++s) dVT[j] = mfma16(op[s * FP + 16 * j], pd[s], dVT[j]);
            const float* qp = Qt + (4 * lq) * EPQ + lr;
#pragma unroll
            for (int j = 0; j < NET; ++j)
#pragma unroll
                for (int s = 0; s < 4; ++s) dKT[j] = mfma16(qp[s * EPQ + 16 * j], ds[s], dKT[j]);
        }
        // dQ[l, e] += sum_tok dS[l, tok] K[tok, e]: transpose this wave's dS through its private LDS scratch
#pragma unroll
        for (int r = 0; r < 4; ++r) myscr[(4 * lq + r) * SCR + lr] = ds[r];
        {
            float aT[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) aT[s] = myscr[lr * SCR + 4 * s + lq];
            float* dst = dQs + buf * BWD_BL * DQW + (4 * lq) * DQW + lr;
            const float* kq = Kw + lq * EPQ + lr;  // K[tok0 + 4s + lq][16j + lr]
#pragma unroll
            for (int j = 0; j < NET; ++j) {
                f32x4 q = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < 4; ++s) q = mfma16(aT[s], kq[4 * s * EPQ + 16 * j], q);
#pragma unroll
                for (int r = 0; r < 4; ++r) atomicAdd(dst + r * DQW + 16 * j, q[r]);  // ds_add_f32 across the waves
            }
        }
        if (more) commit(buf ^ 1);
        __syncthreads();
    }
    // flush the last step's dQ partials
    if (step_beg < step_end) {
        const int buf = (step_end - 1 - step_beg) & 1;
        const float* src = dQs + buf * BWD_BL * DQW;
        const int l0 = (step_end - 1) * BWD_BL;
        for (int i = tid; i < BWD_BL * a.E; i += NT) {
            int r = i / a.E, c = i % a.E;
            if (l0 + r < a.L) atomicAdd(a.dQ + (long long)(l0 + r) * a.E + c, src[r * DQW + c]);
        }
    }
    __syncthreads();
    // epilogue: dK^T / dV^T tiles -> token-major rows through per-wave LDS staging, row-contiguous atomics
    {
        float* st = sm + wave * 16 * BWD_STW;
        flush_tiles<NET>(dKT, st, lane, tok0, a.Tp, a.E, a.dPX + (long long)b * a.Tp * a.E, a.E);
        flush_tiles<NFT>(dVT, st, lane, tok0, a.Tp, a.F, a.dX + (long long)b * a.Tp * a.F, a.F);
    }
}

template <int ES, int NFT>
static int launch_fwd(const AttnFwdArgs& a, hipStream_t st) {
    constexpr int EP = 4 * ES + 2, FP = 16 * NFT + 4;
    size_t smem = (size_t)FWD_BT * (EP + FP) * 4;
    const int nLT = cdiv(a.L, 16 * FWD_WAVES);
    hipLaunchKernelGGL((label_attn_fwd_kernel<ES, NFT>), dim3(a.B * nLT), dim3(FWD_WAVES * 64), smem, st, a);
    GZSL_CHECK_LAUNCH();
    return 0;
}

template <int ES, int NFT>
static int launch_bwd(const AttnBwdArgs& a, hipStream_t st) {
    // 8 waves (128 tokens) per workgroup while the keys fit LDS beside the streamed tiles; 4 waves for E > 100
    constexpr int NW = (ES <= 25) ? 8 : 4;
    using P = BwdLds<ES, NFT, NW>;
    static_assert(P::floats * 4 <= 160 * 1024, "backward LDS plan exceeds 160 KiB");
    const int nTT = cdiv(a.Tp, 16 * NW);
    hipLaunchKernelGGL((label_attn_bwd_kernel<ES, NFT, NW>), dim3(a.B * nTT * a.nsplit), dim3(NW * 64),
                       (size_t)P::floats * 4, st, a);
    GZSL_CHECK_LAUNCH();
    return 0;
}

// shape classes the kernels are instantiated for: E <= 8 / 100 / 200 keys, F <= 32 / 208 values
#define GZ_DISPATCH(FN, a, st)                                   \
    do {                                                         \
        if (a.F <= 32) {                                         \
            if (a.E <= 8) return FN<2, 2>(a, st);                \
            if (a.E <= 100) return FN<25, 2>(a, st);             \
            if (a.E <= 200) return FN<50, 2>(a, st);             \
        } else if (a.F <= 208) {                                 \
            if (a.E <= 8) return FN<2, 13>(a, st);               \
            if (a.E <= 100) return FN<25, 13>(a, st);            \
            if (a.E <= 200) return FN<50, 13>(a, st);            \
        }                                                        \
        return GZSL_EINVAL;                                      \
    } while (0)

extern "C" int gzsl_label_attn_fwd(const float* PX, const float* X, const float* Q, float* O, float* LSE, int B, int Tp,
                                   int E, int F, int L, int l_offset, int L_total, uint32_t seed, uint32_t drop_thresh,
                                   float drop_scale, void* stream) {
    if (B <= 0 || L <= 0) return 0;
    if (Tp <= 0 || E <= 0 || F <= 0) return GZSL_EINVAL;
    AttnFwdArgs a{PX, X, Q, O, LSE, B, Tp, E, F, L, l_offset, L_total, seed, drop_thresh, drop_scale};
    hipStream_t st = (hipStream_t)stream;
    GZ_DISPATCH(launch_fwd, a, st);
}

extern "C" int gzsl_attn_delta(const float* dO, const float* O, float* delta, long long rows, int F, void* stream) {
    if (rows <= 0) return 0;
    int blocks = (int)((rows + 3) / 4);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(attn_delta_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dO, O, delta, rows, F);
    GZSL_CHECK_LAUNCH();
    return 0;
}

extern "C" int gzsl_label_attn_bwd(const float* dO, const float* PX, const float* X, const float* Q, const float* LSE,
                                   const float* DELTA, float* dPX, float* dX, float* dQ, int B, int Tp, int E, int F,
                                   int L, int l_offset, int L_total, int nsplit, uint32_t seed, uint32_t drop_thresh,
                                   float drop_scale, void* stream) {
    if (B <= 0 || L <= 0) return 0;
    if (Tp <= 0 || E <= 0 || F <= 0 || nsplit <= 0) return GZSL_EINVAL;
    AttnBwdArgs a{dO, PX, X, Q, LSE, DELTA, dPX, dX, dQ, B, Tp, E, F, L, l_offset, L_total, nsplit, seed, drop_thresh,
                  drop_scale};
    hipStream_t st = (hipStream_t)stream;
    GZ_DISPATCH(launch_bwd, a, st);
}
