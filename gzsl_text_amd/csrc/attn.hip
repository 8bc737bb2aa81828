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
#include <type_traits>
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

// LDS plan of the forward kernel (shared with its launcher).  Both operand tiles are read 16 bytes per lane:
//   keys   Ks[32 tokens][EP]: lane (token lr, quarter lq) reads e = 16g + 4lq .. +3, its k values for four consecutive
//          MFMA steps of S^T = K.Q^T (the order of the e sum is free as long as the Q fragment agrees);
//   values Vs[32 tokens][FP]: accumulator tile j of lane row i holds f = fwd_f(j, i): tiles are grouped in fours
//          (f = 64*(j/4) + 4*i + j%4), then a pair, then a single tile, so that one ds_read_b128 at token row
//          4lq + sp holds the A operands of FOUR tiles for k step sp.
// ds_read_b128 serves lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31} (+32), banks (a/4) mod 64
// (MI355X_MICROARCH.md, LDS): conflict-free when EP = 8 (mod 16) and FP = 0 (mod 16).
template <int ES, int NFT>
struct FwdLds {
    static constexpr int E4 = 4 * ES;
    static constexpr int EP = (E4 % 16 == 8) ? E4 : ((E4 + 7) / 16 * 16 + 8);
    static constexpr int FP = 16 * NFT;
    static constexpr int NG = ES / 4;   // whole 4-step groups over e
    static constexpr int REM = ES % 4;  // leftover steps: e = 16*NG + 4*j + lq
    static constexpr int G4 = NFT / 4 * 4;
    static constexpr bool HAS2 = (NFT - G4) >= 2;
    static constexpr bool HAS1 = ((NFT - G4) & 1) != 0;
    static constexpr int floats = FWD_BT * (EP + FP);
};

template <int NFT>
__device__ __forceinline__ int fwd_f(int j, int i) {
    constexpr int G4 = NFT / 4 * 4;
    constexpr bool HAS2 = (NFT - G4) >= 2;
    if (j < G4) return 64 * (j >> 2) + 4 * i + (j & 3);
    if (HAS2 && j < G4 + 2) return 16 * G4 + 2 * i + (j - G4);
    return 16 * (G4 + (HAS2 ? 2 : 0)) + i;
}

// G = floats per staging load (4 when E and F are multiples of 4: 16-byte loads of the contiguous 32-row tile).
template <int ES, int NFT, int G, bool DROP>
__global__ __launch_bounds__(FWD_WAVES * 64, 2) void label_attn_fwd_kernel(AttnFwdArgs a) {
    using P = FwdLds<ES, NFT>;
    constexpr int EP = P::EP, FP = P::FP, NG = P::NG, REM = P::REM, G4 = P::G4;
    constexpr bool HAS2 = P::HAS2, HAS1 = P::HAS1;
    constexpr int NT = FWD_WAVES * 64;
    constexpr int NKL = (FWD_BT * 4 * ES / G + NT - 1) / NT;   // staging loads per thread (keys)
    constexpr int NVL = (FWD_BT * 16 * NFT / G + NT - 1) / NT;  // (values)
    typedef float vecG __attribute__((ext_vector_type(G)));
    typedef float f32x2 __attribute__((ext_vector_type(2)));
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

    // Q^T fragment, B operand of S^T = K.Q^T, in the k order of the key reads:
    //   qreg[4g + j] = Q[lbl][16g + 4lq + j] (whole groups), qreg[4NG + j] = Q[lbl][16NG + 4j + lq] (leftover steps)
    // (clamped address + multiply by 0/1: no branches around loads)
    float qreg[ES];
    {
        const float* qrow = a.Q + (long long)min(lbl, a.L - 1) * a.E;
#pragma unroll
        for (int s = 0; s < ES; ++s) {
            if (G == 4 && s < 4 * NG) {  // E % 4 == 0: the four steps of a group are one aligned vector
                if ((s & 3) == 0) {
                    const int e = 16 * (s >> 2) + 4 * lq;
                    const f32x4 v = *reinterpret_cast<const f32x4*>(qrow + min(e, a.E - 4));
                    const float keep = (lbl_ok && e < a.E) ? 1.f : 0.f;
#pragma unroll
                    for (int j = 0; j < 4; ++j) qreg[s + j] = v[j] * keep;
                }
                continue;
            }
            const int e = (s < 4 * NG) ? 16 * (s >> 2) + 4 * lq + (s & 3) : 16 * NG + 4 * (s - 4 * NG) + lq;
            qreg[s] = qrow[min(e, a.E - 1)] * ((lbl_ok && e < a.E) ? 1.f : 0.f);
        }
    }
    // zero both tiles once: pad columns are never written by the staging copy and must not hold NaN patterns
    static_assert(P::floats % 4 == 0, "LDS plan is a whole number of 16-byte words");
    for (int i = tid * 4; i < P::floats; i += NT * 4) *reinterpret_cast<f32x4*>(Ks + i) = f32x4{0.f, 0.f, 0.f, 0.f};

    // the 32-row tile of keys (values) is one contiguous block of 32*E (32*F) floats: copy it flat, G floats per
    // load, and re-stride into the padded LDS rows.  LDS offsets are loop-invariant: computed once.
    int kofs[NKL], vofs[NVL];
#pragma unroll
    for (int i = 0; i < NKL; ++i) {
        const int f = (tid + NT * i) * G;
        kofs[i] = (f < FWD_BT * a.E) ? (f / a.E) * EP + f % a.E : -1;
    }
#pragma unroll
    for (int i = 0; i < NVL; ++i) {
        const int f = (tid + NT * i) * G;
        vofs[i] = (f < FWD_BT * a.F) ? (f / a.F) * FP + f % a.F : -1;
    }

    const uint32_t rowkey = gz_row_key(a.seed, GZSL_STREAM_ATTN, (uint32_t)b * (uint32_t)a.L_total + (uint32_t)(a.l_offset + lbl));
    uint32_t colmul[4];  // (4lq + r) * C: the lane's share of the dropout hash's column product
#pragma unroll
    for (int r = 0; r < 4; ++r) colmul[r] = (uint32_t)(4 * lq + r) * 0x85EBCA6Bu;
    f32x4 acc[NFT];
#pragma unroll
    for (int j = 0; j < NFT; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m = -INFINITY, lsum = 0.f;

    const float* PXb = a.PX + (long long)b * a.Tp * a.E;
    const float* Xb = a.X + (long long)b * a.Tp * a.F;

    vecG kreg[NKL], vreg[NVL];
    // global -> registers; stays in flight while the previous tile is computed.  Clamped addresses, no validity
    // selects (a `cond ? load : 0` makes hipcc branch around the load and wait inside the branch): rows past Tp hold
    // finite neighbouring data whose scores are masked to -inf below, so their P is exactly 0.
    // element offsets of this thread's loads inside a FULL tile: loop-invariant, so the steady state is a wave-uniform
    // base (scalar registers) + one 32-bit lane offset per load, no per-tile address arithmetic on the vector ALU
    unsigned kgo[NKL], vgo[NVL];
#pragma unroll
    for (int i = 0; i < NKL; ++i) kgo[i] = (unsigned)min((tid + NT * i) * G, FWD_BT * a.E - G);
#pragma unroll
    for (int i = 0; i < NVL; ++i) vgo[i] = (unsigned)min((tid + NT * i) * G, FWD_BT * a.F - G);
    auto issue = [&](int t0) {
        const float* kb = PXb + (long long)t0 * a.E;
        const float* vb = Xb + (long long)t0 * a.F;
        if (t0 + FWD_BT <= a.Tp) {
#pragma unroll
            for (int i = 0; i < NKL; ++i) kreg[i] = *reinterpret_cast<const vecG*>(kb + kgo[i]);
#pragma unroll
            for (int i = 0; i < NVL; ++i) vreg[i] = *reinterpret_cast<const vecG*>(vb + vgo[i]);
        } else {  // last, partial tile of the note
            const int rows = a.Tp - t0;
            const int nk = rows * a.E, nv = rows * a.F;
#pragma unroll
            for (int i = 0; i < NKL; ++i) kreg[i] = *reinterpret_cast<const vecG*>(kb + min((tid + NT * i) * G, nk - G));
#pragma unroll
            for (int i = 0; i < NVL; ++i) vreg[i] = *reinterpret_cast<const vecG*>(vb + min((tid + NT * i) * G, nv - G));
        }
    };
    auto commit = [&]() {  // registers -> LDS (one 16-byte store per load when G == 4)
#pragma unroll
        for (int i = 0; i < NKL; ++i)
            if (kofs[i] >= 0) *reinterpret_cast<vecG*>(Ks + kofs[i]) = kreg[i];
#pragma unroll
        for (int i = 0; i < NVL; ++i)
            if (vofs[i] >= 0) *reinterpret_cast<vecG*>(Vs + vofs[i]) = vreg[i];
    };

    issue(0);
    for (int t0 = 0; t0 < a.Tp; t0 += FWD_BT) {
        __syncthreads();  // previous tile fully consumed (first pass: zero fill done)
        commit();
        __syncthreads();
        if (t0 + FWD_BT < a.Tp) issue(t0 + FWD_BT);
#pragma unroll
        for (int u = 0; u < FWD_BT / 16; ++u) {
            // S^T sub-tile: rows = tokens t0+16u+4lq+r, cols = labels
            f32x4 s0 = f32x4{0.f, 0.f, 0.f, 0.f}, s1 = s0;
            const float* kp = Ks + (16 * u + lr) * EP;
            // key groups are read one group ahead of their MFMAs and pinned there (otherwise every group of four
            // MFMAs waits out the LDS latency of its own read)
            float kr[REM > 0 ? REM : 1];
#pragma unroll
            for (int j = 0; j < REM; ++j) kr[j] = kp[16 * NG + 4 * j + lq];
            f32x4 kf[2];
            if (NG > 0) kf[0] = *reinterpret_cast<const f32x4*>(kp + 4 * lq);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                if (g + 1 < NG) kf[(g + 1) & 1] = *reinterpret_cast<const f32x4*>(kp + 16 * (g + 1) + 4 * lq);
                __builtin_amdgcn_sched_barrier(0);
                const f32x4 k4 = kf[g & 1];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (j & 1) s1 = mfma16(k4[j], qreg[4 * g + j], s1);
                    else s0 = mfma16(k4[j], qreg[4 * g + j], s0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int j = 0; j < REM; ++j) {
                if (j & 1) s1 = mfma16(kr[j], qreg[4 * NG + j], s1);
                else s0 = mfma16(kr[j], qreg[4 * NG + j], s0);
            }
            float sv[4];
            float mt = -INFINITY;
            const int tbase = t0 + 16 * u + 4 * lq;
#pragma unroll
            for (int r = 0; r < 4; ++r) sv[r] = s0[r] + s1[r];
            if (t0 + FWD_BT > a.Tp) {  // block-uniform: only the last tile has rows past the end of the note
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (tbase + r >= a.Tp) sv[r] = -INFINITY;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) mt = fmaxf(mt, sv[r]);
            // m is uniform over the four lanes of a label column, so "some lane's own maximum exceeds m + THR" is the
            // same wave-uniform condition as for the column maximum: the two cross-lane exchanges are only paid when
            // a rescale actually happens (always on the first tile, m = -inf; rarely afterwards)
            if (__any(mt > m + RESCALE_THR)) {
                mt = fmaxf(mt, __shfl_xor(mt, 16));
                mt = fmaxf(mt, __shfl_xor(mt, 32));
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
                if (DROP) {  // compile-time: a runtime test here splits the loop body into many basic blocks
                    // gz_u32(rowkey, tbase + r) with the column product built from a lane constant and a wave-uniform
                    // term (scalar ALU) instead of a quarter-rate vector multiply per element
                    const uint32_t uu = gz_mix32(rowkey ^ (colmul[r] + (uint32_t)(t0 + 16 * u) * 0x85EBCA6Bu));
                    // (folding the keep scale into the final 1/lsum instead measured 2.5 % slower)
                    p[r] = (uu >= a.thresh) ? p[r] * a.scale : 0.f;
                }
            }
            // O^T[f, label] += sum_t V[t, f] * P[t, label]; k step sp pairs tokens {4q + sp}.  sp outermost:
            // consecutive MFMAs hit different accumulators (the 16x16x4 form has a 40-cycle dependent latency)
            const float* vp = Vs + (16 * u + 4 * lq) * FP;
            float vf[2][NFT];  // A fragments of k step sp: one float per accumulator tile, read one k step ahead
            auto ldv = [&](int sp, int slot) {
                const float* vrow = vp + sp * FP;
#pragma unroll
                for (int gi = 0; gi < NFT / 4; ++gi) {
                    const f32x4 v4 = *reinterpret_cast<const f32x4*>(vrow + 64 * gi + 4 * lr);
#pragma unroll
                    for (int e = 0; e < 4; ++e) vf[slot][4 * gi + e] = v4[e];
                }
                if (HAS2) {
                    const f32x2 v2 = *reinterpret_cast<const f32x2*>(vrow + 16 * G4 + 2 * lr);
                    vf[slot][G4] = v2[0];
                    vf[slot][G4 + 1] = v2[1];
                }
                if (HAS1) vf[slot][NFT - 1] = vrow[16 * (G4 + (HAS2 ? 2 : 0)) + lr];
            };
            ldv(0, 0);
#pragma unroll
            for (int sp = 0; sp < 4; ++sp) {
                if (sp + 1 < 4) ldv(sp + 1, (sp + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < NFT; ++j) acc[j] = mfma16(vf[sp & 1][j], p[sp], acc[j]);
                __builtin_amdgcn_sched_barrier(0);
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
        // acc[j][r] = O[lbl][fwd_f(j, 4lq + r)]: four tiles of a group give four consecutive f per (lq, r)
#pragma unroll
        for (int gi = 0; gi < NFT / 4; ++gi) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int f = 64 * gi + 16 * lq + 4 * r;
                if (vec && f + 3 < a.F) {
                    *reinterpret_cast<float4*>(Op + f) = make_float4(acc[4 * gi][r] * inv, acc[4 * gi + 1][r] * inv,
                                                                     acc[4 * gi + 2][r] * inv, acc[4 * gi + 3][r] * inv);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (f + e < a.F) Op[f + e] = acc[4 * gi + e][r] * inv;
                }
            }
        }
        if (HAS2) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int f = fwd_f<NFT>(G4 + e, 4 * lq + r);
                    if (f < a.F) Op[f] = acc[G4 + e][r] * inv;
                }
        }
        if (HAS1) {
            const int f = fwd_f<NFT>(NFT - 1, 4 * lq);
            if (vec && f + 3 < a.F) {
                *reinterpret_cast<float4*>(Op + f) = make_float4(acc[NFT - 1][0] * inv, acc[NFT - 1][1] * inv,
                                                                 acc[NFT - 1][2] * inv, acc[NFT - 1][3] * inv);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (f + r < a.F) Op[f + r] = acc[NFT - 1][r] * inv;
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
    if ((F & 3) == 0 && (((uintptr_t)dO | (uintptr_t)O) & 15) == 0) {  // 16-byte loads: a lane owns four columns
        const int F4 = F >> 2;
        const float4* a4 = reinterpret_cast<const float4*>(dO);
        const float4* b4 = reinterpret_cast<const float4*>(O);
        for (; row < rows; row += stride) {
            float s = 0.f;
            for (int f = lane; f < F4; f += 64) {
                const float4 a = a4[row * F4 + f], b = b4[row * F4 + f];
                s += a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
            }
            s = wave_sum(s);
            if (lane == 0) delta[row] = s;
        }
        return;
    }
    for (; row < rows; row += stride) {
        float s = 0.f;
        for (int f = lane; f < F; f += 64) s += dO[row * F + f] * O[row * F + f];
        s = wave_sum(s);
        if (lane == 0) delta[row] = s;
    }
}

#define BWD_BL 16  // labels per pipeline step
#ifndef GZ_ABL
#define GZ_ABL 0  // diagnostic builds only (tools/ablate_bwd.py): bit 0 no dQ, 1 no dK/dV MFMAs, 2 no RNG, 3 no streaming, 4 no epilogue, 5 no S/dP
#endif

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
    // deterministic mode (DET kernels): partial sums leave as plain stores into slabs that gzsl_label_attn_bwd_det then
    // adds up in a fixed order -- dq_slab [B * nTT][L][E] (one slab per (note, token tile)), dpx_slab [nsplit][B][Tp][E],
    // dx_slab [nsplit][B][Tp][F] (one slab per label split)
    float* dq_slab;
    float* dpx_slab;
    float* dx_slab;
};

// column of accumulator tile j, lane row/col i, in a block of NT 16-wide tiles read as vectors along the columns:
// groups of four tiles (col = 64*(j/4) + 4*i + j%4), then a pair (2*i + e), then a single natural tile.
template <int NT>
__device__ __forceinline__ int grp_col(int j, int i) {
    constexpr int G4 = NT / 4 * 4;
    constexpr bool HAS2 = (NT - G4) >= 2;
    if (j < G4) return 64 * (j >> 2) + 4 * i + (j & 3);
    if (HAS2 && j < G4 + 2) return 16 * G4 + 2 * i + (j - G4);
    return 16 * (G4 + (HAS2 ? 2 : 0)) + i;
}

// LDS plan of the backward kernel, shared by the kernel and its launcher.
//
// The streamed 16-label tiles of Q [16][E] and dO [16][F] and the workgroup's keys K [16*NW][E] are kept as ONE natural
// row-major image each, row stride RS = 16 * odd, with the 16-byte chunk index XORed with (row >> 1) & 3.  With that
// stride and XOR both ways of reading an image are conflict-free ds_read_b128 (4 LDS cycles; lane groups
// {0-3,12-15,20-27}, {4-11,16-19,28-31} (+32), banks (a/4) mod 64: MI355X_MICROARCH.md, LDS):
//   "along k"   : lane (row lr, quarter lq) reads columns 16g + 4lq .. +3 = its k values for four consecutive MFMA steps
//                 (A operands of S = Q.K^T and dP = dO.V^T, B operand K of S);
//   "along rows": lane (lr, lq) reads row 4lq + r, columns grp_col(4jg.., lr) = the B operands of FOUR accumulator tiles
//                 for k step r (dK += dS^T.Q and dV += Pd^T.dO, whose A operands dS / Pd are the S-tile registers).
// Nothing is stored twice and nothing is transposed through LDS except dS, once per step, for dQ.
template <int ES, int NFT, int NW, bool PAIR = false>
struct BwdLds {
    static constexpr int NWV = PAIR ? 2 * NW : NW;  // waves in the workgroup (PAIR: a K-role and a V-role wave per token group)
    static constexpr int NET = (4 * ES + 15) / 16;  // 16-wide tiles over E
    static constexpr int NG = ES / 4;               // whole 4-step groups over e (S)
    static constexpr int REM = ES % 4;              // leftover steps of S: e = 16*NG + 4*j + lq
    static constexpr int FS = 4 * NFT;              // k steps over F (dP): whole groups, zero-padded values
    static constexpr int rs(int w) { return ((w + 15) / 16) % 2 == 1 ? (w + 15) / 16 * 16 : (w + 15) / 16 * 16 + 16; }
    static constexpr int RSQ = rs(16 * NET);        // row stride of the Q / K images
    static constexpr int RSO = rs(16 * NFT);        // row stride of the dO image
    static constexpr int ks(int steps) { return (steps % 8 == 4) ? steps : steps + 4; }
    static constexpr int KSD = ks(4 * NW);          // dS image ("p" order: elem(l, tok) at l*RSD + (tok&3)*KSD + (tok>>2))
    static constexpr int RSD = 4 * KSD + 8;
    static constexpr int DQW = 16 * NET;
    static constexpr int TQ = BWD_BL * RSQ, TO = BWD_BL * RSO, TSZ = TQ + TO;
    static constexpr int oTiles = 0;                      // [2][TQ + TO]    Q and dO tiles, double-buffered
    static constexpr int oAux = oTiles + 2 * TSZ;         // [2][48]         lse[16], delta[16], dropout row keys[16]
    static constexpr int oDS = oAux + 2 * 48;             // [2][16][RSD]    dS of the workgroup's tokens
    static constexpr int odQ = oDS + 2 * BWD_BL * RSD;    // [2][16][DQW]    dQ tiles awaiting their flush
    static constexpr int oKp = odQ + 2 * BWD_BL * DQW;    // [NW*16][RSQ]    this workgroup's keys
    static constexpr int oXch = oKp + NW * 16 * RSQ;      // [2][NW][64][4]  PAIR only: Pd and dP tiles exchanged by a wave pair
    static constexpr int total = oXch + (PAIR ? 2 * NW * 256 : 0);
    static constexpr int STW = 129;                       // epilogue staging row: 128 columns per pass, odd stride
    static constexpr int stage = NWV * 16 * STW;          // (aliases the front)
    static constexpr int floats = total > stage ? total : stage;
};

__device__ __forceinline__ int swz(int row) { return (row >> 1) & 3; }
// float offset of element (row, col) in a swizzled natural image of row stride RS
__device__ __forceinline__ int img_off(int row, int col, int RS) { return row * RS + 4 * ((col >> 2) ^ swz(row)) + (col & 3); }

// Accumulator tiles hold [token 4*lq + r][column grp_col(j, lr)].  Stage 128 columns at a time as [token][column] in the
// wave's private LDS rows and add whole token rows to global memory (256 contiguous bytes per wave-instruction: the
// shape float atomics run at full rate in).  All indices are compile-time after unrolling (no scratch).
template <int NTILES, int STW, bool DET, int NARR>
__device__ __forceinline__ void flush_rows(const f32x4 (&tiles)[NARR], float* st, int lane, int tok0, int Tp,
                                           int width, float* out, int ld) {
    const int lr = lane & 15, lq = lane >> 4;
    constexpr int NPASS = (16 * NTILES + 127) / 128;
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
#pragma unroll
        for (int j = 0; j < NTILES; ++j) {
            if (grp_col<NTILES>(j, 0) / 128 == ps) {
                const int c = grp_col<NTILES>(j, lr) - 128 * ps;
#pragma unroll
                for (int r = 0; r < 4; ++r) st[(4 * lq + r) * STW + c] = tiles[j][r];
            }
        }
        wave_lds_fence();  // the rows read below were staged by other lanes of this wave
        const int ncol = min(128, 16 * NTILES - 128 * ps);
        for (int trow = 0; trow < 16; ++trow) {
            const int t = tok0 + trow;
            if (t >= Tp) break;
            for (int c = lane; c < ncol; c += 64) {
                const int col = 128 * ps + c;
                if (col < width) {
                    if (DET) out[(long long)t * ld + col] = st[trow * STW + c];  // this workgroup's own slab rows
                    else atomicAdd(out + (long long)t * ld + col, st[trow * STW + c]);
                }
            }
        }
        wave_lds_fence();  // ... and the next pass overwrites them
    }
}

// G = floats per staging load (4 when E and F are multiples of 4)
//
// PAIR (E > 100): a token's state is K, V, dK, dV = 4(E+F) floats; with one wave owning all of it a 64-token workgroup
// fills the register file at ONE wave per SIMD.  Instead two waves share each group of 16 tokens: the K-role wave
// computes S = Q.K^T, P, the dropout mask and dK, the V-role wave dP = dO.V^T and dV.  They exchange the 16x16 Pd and
// dP tiles (same lane layout: one 16-byte LDS store and load per lane) across one extra barrier per step; each
// holds ONE accumulator array, so the workgroup runs 8 waves = two per SIMD.
// DROP: dropout on (compile-time: a runtime test splits the softmax block into one basic block per score element).
// (Measured and rejected: letting waves 4.. run the previous step's dQ tile product at the START of their step, so that
// the two waves of a SIMD are not in the same phase at the same time, cost +5.6 % at E = 100 and +7.5 % at E = 200.)
template <int ES, int NFT, int NW, int G, bool PAIR, bool DROP, bool DET>
__global__ __launch_bounds__((PAIR ? 2 : 1) * NW * 64) void label_attn_bwd_kernel(AttnBwdArgs a) {
    using P = BwdLds<ES, NFT, NW, PAIR>;
    constexpr int NWV = P::NWV;
    constexpr int NET = P::NET, NG = P::NG, REM = P::REM, FS = P::FS, RSQ = P::RSQ, RSO = P::RSO, DQW = P::DQW;
    constexpr int RSD = P::RSD, KSD = P::KSD, TQ = P::TQ, TSZ = P::TSZ;
    constexpr int NT = NWV * 64;
    constexpr int WQ = 4 * ES, WO = 16 * NFT;  // staged tile widths (compile-time; columns past E / F are clamped copies)
    typedef float vecG __attribute__((ext_vector_type(G)));
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* sm = reinterpret_cast<float*>(smem_raw);
    float* tiles = sm + P::oTiles;
    float* aux = sm + P::oAux;
    float* dSs = sm + P::oDS;
    float* dQs = sm + P::odQ;
    float* Kp = sm + P::oKp;

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wave = PAIR ? (wv & (NW - 1)) : wv;   // token group of this wave
    const bool role_v = PAIR && wv >= NW;           // PAIR: waves [NW, 2NW) hold V / dV, waves [0, NW) hold K / dK
    const bool do_k = !PAIR || !role_v, do_v = !PAIR || role_v;
    static_assert(!PAIR || (NW & (NW - 1)) == 0, "PAIR needs a power-of-two token-wave count");
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

    // zero all of LDS once: pad words of the staged images must never hold NaN bit patterns
    static_assert(P::floats % 4 == 0, "LDS plan is a whole number of 16-byte words");
    for (int i = tid * 4; i < P::floats; i += NT * 4) *reinterpret_cast<f32x4*>(sm + i) = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    // this workgroup's keys; batched straight-line loads from clamped addresses, columns past E stay zero
    if constexpr (G == 4) {  // E % 4 == 0, 16-byte aligned rows: one vector per four columns (a swizzle chunk)
        constexpr int KPT = (NW * 16 * (WQ / 4) + NT - 1) / NT;
        f32x4 kk[KPT];
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            const int idx = tid + NT * i, r = idx / (WQ / 4), c = 4 * (idx % (WQ / 4));
            kk[i] = *reinterpret_cast<const f32x4*>(PXb + (long long)min(tt * 16 * NW + r, a.Tp - 1) * a.E + min(c, a.E - 4));
        }
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            const int idx = tid + NT * i, r = idx / (WQ / 4), c = 4 * (idx % (WQ / 4));
            if (r < NW * 16 && tt * 16 * NW + r < a.Tp && c < a.E) *reinterpret_cast<f32x4*>(Kp + img_off(r, c, RSQ)) = kk[i];
        }
    } else {
        constexpr int KPT = (NW * 16 * WQ + NT - 1) / NT;
        float kk[KPT];
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            const int idx = tid + NT * i, r = idx / WQ, c = idx % WQ;
            kk[i] = PXb[(long long)min(tt * 16 * NW + r, a.Tp - 1) * a.E + min(c, a.E - 1)];
        }
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            const int idx = tid + NT * i, r = idx / WQ, c = idx % WQ;
            if (r < NW * 16 && tt * 16 * NW + r < a.Tp && c < a.E) Kp[img_off(r, c, RSQ)] = kk[i];
        }
    }
    // values stay in registers, B operand of dP = dO.V^T in the k order of the "along k" reads:
    //   vT[4g + j] = V[tok][16g + 4lq + j]   (mask by multiply: loads stay unconditional)
    float vT[FS];
    if (do_v) {
        const float* vrow = Xb + (long long)min(tok, a.Tp - 1) * a.F;
        if constexpr (G == 4) {  // F % 4 == 0: the four columns of a group are one aligned vector
#pragma unroll
            for (int g4 = 0; g4 < FS / 4; ++g4) {
                const int f = 16 * g4 + 4 * lq;
                const f32x4 v = *reinterpret_cast<const f32x4*>(vrow + min(f, a.F - 4));
                const float keep = (tok_ok && f < a.F) ? 1.f : 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) vT[4 * g4 + e] = v[e] * keep;
            }
        } else {
#pragma unroll
            for (int s = 0; s < FS; ++s) {
                const int f = 16 * (s >> 2) + 4 * lq + (s & 3);
                vT[s] = vrow[min(f, a.F - 1)] * ((tok_ok && f < a.F) ? 1.f : 0.f);
            }
        }
    }

    // [token 4lq + r][column grp_col(j, lr)].  PAIR: one array per wave, dK tiles in the K-role wave, dV tiles in the other
    constexpr int NACC = NET > NFT ? NET : NFT;
    f32x4 dK[PAIR ? 1 : NET], dV[PAIR ? 1 : NFT], acc[PAIR ? NACC : 1];
#pragma unroll
    for (int j = 0; j < (PAIR ? 1 : NET); ++j) dK[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < (PAIR ? 1 : NFT); ++j) dV[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < (PAIR ? NACC : 1); ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const uint32_t seedmix = gz_mix32(a.seed + GZSL_STREAM_ATTN * 0x632BE5ABu);
    const uint32_t rowbase = (uint32_t)b * (uint32_t)a.L_total + (uint32_t)a.l_offset;
    const uint32_t tokmul = (uint32_t)tok * 0x85EBCA6Bu;  // gz_u32(key, tok) = mix32(key ^ tokmul)

    // staging: thread -> G-float pieces of the 16 x WQ and 16 x WO tiles; offsets are loop-invariant
    constexpr int QPT = (BWD_BL * WQ / G + NT - 1) / NT;
    constexpr int OPT = (BWD_BL * WO / G + NT - 1) / NT;
    constexpr int DQF = (BWD_BL * DQW + NT - 1) / NT;
    // global offsets are (first label of the step) * width [wave-uniform: scalar ALU] + a per-slot constant, clamped to
    // the slot's position in the LAST label row; nothing is multiplied per step (VALU work is paid in matrix time)
    int q_off[QPT], q_lim[QPT], q_lds[QPT], o_off[OPT], o_lim[OPT], o_lds[OPT];
#pragma unroll
    for (int i = 0; i < QPT; ++i) {
        const int f = (tid + NT * i) * G, r = f / WQ, c = min(f % WQ, a.E - G);
        q_off[i] = r * a.E + c;
        q_lim[i] = (a.L - 1) * a.E + c;
        q_lds[i] = (r < BWD_BL) ? img_off(r, f % WQ, RSQ) : -1;
    }
#pragma unroll
    for (int i = 0; i < OPT; ++i) {
        const int f = (tid + NT * i) * G, r = f / WO, c = min(f % WO, a.F - G);
        o_off[i] = r * a.F + c;
        o_lim[i] = (a.L - 1) * a.F + c;
        o_lds[i] = (r < BWD_BL) ? img_off(r, f % WO, RSO) : -1;
    }
    vecG pq[QPT], po[OPT];
    float plse = 0.f, pdel = 0.f, plinf = 0.f;
    uint32_t pkey = 0;
    // Straight-line loads from CLAMPED addresses, no validity selects (a `cond ? load : 0` makes hipcc branch around
    // each load and wait vmcnt(0) inside the branch).  Rows past L are neutralised through lse = +inf (P = 0, so
    // dS = Pd = 0 whatever finite Q / dO row was read); columns past E / F meet zero key / value pads, or produce
    // accumulator columns that are never stored.
    const float* dOb = a.dO + (long long)b * a.L * a.F;   // in-note offsets fit 32 bits (L*F, L*E < 2^31: checked on the host)
    const float* lseb = a.LSE + (long long)b * a.L;
    const float* delb = a.DELTA + (long long)b * a.L;
    auto prefetch = [&](int step) {
        const int l0 = step * BWD_BL;
        const int qb = l0 * a.E, ob = l0 * a.F;
#pragma unroll
        for (int i = 0; i < QPT; ++i)
            pq[i] = *reinterpret_cast<const vecG*>(a.Q + (unsigned)min(qb + q_off[i], q_lim[i]));
#pragma unroll
        for (int i = 0; i < OPT; ++i)
            po[i] = *reinterpret_cast<const vecG*>(dOb + (unsigned)min(ob + o_off[i], o_lim[i]));
        if (wv == 0) {  // lse, delta and the dropout row keys of the 16 labels: one wave's job
            const int l = l0 + (tid & (BWD_BL - 1));
            const unsigned o = (unsigned)min(l, a.L - 1);
            plse = lseb[o];
            pdel = delb[o];
            plinf = (l < a.L) ? 0.f : INFINITY;  // added at commit time (nothing may wait on a load here)
            pkey = gz_mix32(seedmix ^ ((rowbase + (uint32_t)l) * 0x9E3779B1u));
        }
    };
    auto commit = [&](int buf) {  // registers -> the tile images of buffer `buf`
        float* tq = tiles + buf * TSZ;
#pragma unroll
        for (int i = 0; i < QPT; ++i)
            if (q_lds[i] >= 0) *reinterpret_cast<vecG*>(tq + q_lds[i]) = pq[i];
#pragma unroll
        for (int i = 0; i < OPT; ++i)
            if (o_lds[i] >= 0) *reinterpret_cast<vecG*>(tq + TQ + o_lds[i]) = po[i];
        if (tid < BWD_BL) {
            float* ax = aux + buf * 48;
            ax[tid] = plse + plinf;  // +inf -> P = 0 for labels past L
            ax[16 + tid] = pdel;
            ax[32 + tid] = __uint_as_float(pkey);
        }
    };

    // Each workgroup starts its label sweep at a different step: workgroups that run together would otherwise all
    // add their dQ partials into the SAME 16 rows at the same time, and same-address float atomics serialise at the
    // memory side (an order of magnitude below the 1.3 TB/s spread-out atomic rate).
    const int nsteps = step_end - step_beg;
    const int rot = nsteps > 0 ? (int)(((unsigned)(tt * 7 + b * 13) * 2654435761u >> 8) % (unsigned)nsteps) : 0;
    auto step_of = [&](int i) { int s = i + rot; return step_beg + (s >= nsteps ? s - nsteps : s); };
    if (nsteps > 0) {
        prefetch(step_of(0));
        commit(0);
    }
    __syncthreads();

    // lane constants of the two read patterns
    const int xk = 4 * (lq ^ swz(lr));  // "along k": chunk (4g + lq) of row lr sits at float 16g + xk
    int xrem[REM > 0 ? REM : 1];        // leftover S steps: column 16NG + 4j + lq of row lr
#pragma unroll
    for (int j = 0; j < REM; ++j) xrem[j] = 16 * NG + 4 * (j ^ swz(lr)) + lq;
    // "along rows": row 4lq + r has swz = (2lq + (r >> 1)) & 3; h = r >> 1
    int x4[2], x2[2], x1[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int sw = (2 * lq + h) & 3;
        x4[h] = 4 * (lr ^ sw);                         // four-tile groups: columns 64jg + 4lr ..
        x2[h] = 4 * ((lr >> 1) ^ sw) + 2 * (lr & 1);   // pair: columns 2lr, 2lr + 1
        x1[h] = 4 * ((lr >> 2) ^ sw) + (lr & 3);       // single tile: column lr
    }
    const float* Kw = Kp + (wave * 16 + lr) * RSQ;  // this lane's key row
    // dQ[l, e] = sum_tok dS[l, tok] K[tok, e] sums over the tokens of ALL waves.  The waves exchange dS through one
    // LDS tile and wave w owns output columns e in [16w, 16w+16) over the whole 16*NW-token K range: no cross-wave
    // reduction (LDS float atomics ran at < 1 lane/clk/CU and cost 3/4 of the kernel).  The product for step i runs
    // inside step i+1 and its flush inside step i+2 (double-buffered dS and dQ tiles): ONE barrier per step.
    auto dq_tile = [&](int buf) {
        const float* dsp = dSs + buf * BWD_BL * RSD;
        float* dqo = dQs + buf * BWD_BL * DQW;
        for (int j = wv; j < NET; j += NWV) {
            f32x4 q0 = f32x4{0.f, 0.f, 0.f, 0.f}, q1 = q0;
            const f32x4* ap = reinterpret_cast<const f32x4*>(dsp + lr * RSD + lq * KSD);  // dS[label lr][tok 4s+lq]
            // K[tok = 4s + lq][e = 16j + lr]: row 4s + lq has swz = (2(s & 1) + (lq >> 1)) & 3
            const float* kq0 = Kp + lq * RSQ + 16 * j + 4 * ((lr >> 2) ^ (lq >> 1)) + (lr & 3);
            const float* kq1 = Kp + lq * RSQ + 16 * j + 4 * ((lr >> 2) ^ (2 + (lq >> 1))) + (lr & 3);
            // operands of token group g (4 k steps) are read one group ahead of their MFMAs and pinned there
            f32x4 af[2];
            float kf[2][4];
            auto ldg = [&](int g, int slot) {
                af[slot] = ap[g];
#pragma unroll
                for (int i = 0; i < 4; ++i) kf[slot][i] = ((i & 1) ? kq1 : kq0)[(4 * (4 * g + i)) * RSQ];
            };
            ldg(0, 0);
#pragma unroll
            for (int g = 0; g < NW; ++g) {
                if (g + 1 < NW) ldg(g + 1, (g + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (i & 1) q1 = mfma16(af[g & 1][i], kf[g & 1][i], q1);
                    else q0 = mfma16(af[g & 1][i], kf[g & 1][i], q0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) dqo[(4 * lq + r) * DQW + 16 * j + lr] = q0[r] + q1[r];
        }
    };
    int dq_goff[DQF], dq_row[DQF];  // per-slot constants of the dQ flush: global offset inside the 16-row block, row
#pragma unroll
    for (int i = 0; i < DQF; ++i) {
        const int idx = tid + NT * i;
        const int r = idx / DQW, c = idx % DQW;
        const bool ok = idx < BWD_BL * DQW && c < a.E;
        dq_goff[i] = r * a.E + c;
        dq_row[i] = ok ? r : (1 << 30);  // never < L - l0
    }
    auto dq_flush = [&](int buf, int l0p) {  // row-contiguous float atomics; fixed trip count, unrolled
        const float* dqo = dQs + buf * BWD_BL * DQW;
        // DET: every (note, token tile) owns a [L][E] slab and each of its label rows is written by exactly one workgroup
        float* dqg = (DET ? a.dq_slab + (long long)(b * nTT + tt) * a.L * a.E : a.dQ) + (long long)l0p * a.E;
        const int nrow = a.L - l0p;
#pragma unroll
        for (int i = 0; i < DQF; ++i) {
            if (dq_row[i] < nrow) {
                if (DET) dqg[dq_goff[i]] = dqo[tid + NT * i];
                else atomicAdd(dqg + dq_goff[i], dqo[tid + NT * i]);
            }
        }
    };

    int l0_p1 = 0, l0_p2 = 0;  // first labels of the previous two steps
    // The step is instantiated twice, once per tile buffer: every LDS address is then (lane constant) + (immediate); with a
    // runtime buffer index the compiler rebuilt ~40 vector addresses per step (vector ALU time is matrix time here).
    auto step_body = [&](auto CURC, const int it) {
        constexpr int cur = decltype(CURC)::value;
        const int l0 = step_of(it) * BWD_BL;
        const bool more = it + 1 < nsteps;
        if (more && !(GZ_ABL & 8)) prefetch(step_of(it + 1));  // in flight during the whole step
        if (it > 1 && !(GZ_ABL & 1)) dq_flush(cur, l0_p2);     // dQ tile of step it-2 (completed before the last barrier)
        const float* tq = tiles + cur * TSZ;
        const float* to = tq + TQ;
        const float* ax = aux + cur * 48;
        // S tile (rows = labels l0 + 4lq + r, cols = tokens) and dP tile; two accumulators each (40-cycle latency)
        f32x4 s0 = f32x4{0.f, 0.f, 0.f, 0.f}, s1 = s0, d0 = s0, d1 = s0;
        if (!(GZ_ABL & 32)) {
            // One flat sequence of items (NG key/query groups, then NFT dO groups), each read one item ahead of its
            // MFMAs and pinned there: the compiler otherwise issues every ds_read right before its use, and each group
            // of four MFMAs then waits out a full LDS latency.
            const float* qp = tq + lr * RSQ;
            const float* op = to + lr * RSO;
            constexpr int NI = NG + NFT;
            f32x4 fa[2], fb[2];
            auto ld = [&](int it, int slot) {
                if (it < NG) {
                    fa[slot] = *reinterpret_cast<const f32x4*>(qp + 16 * it + xk);
                    fb[slot] = *reinterpret_cast<const f32x4*>(Kw + 16 * it + xk);
                } else {
                    fa[slot] = *reinterpret_cast<const f32x4*>(op + 16 * (it - NG) + xk);
                }
            };
            float rq[REM > 0 ? REM : 1], rk[REM > 0 ? REM : 1];
#pragma unroll
            for (int j = 0; j < REM; ++j) {
                rq[j] = qp[xrem[j]];
                rk[j] = Kw[xrem[j]];
            }
            auto run_items = [&](auto I0, auto I1) {
                constexpr int i0 = decltype(I0)::value, i1 = decltype(I1)::value;
                if constexpr (i0 < i1) {
                    ld(i0, i0 & 1);
#pragma unroll
                    for (int it = i0; it < i1; ++it) {
                        if (it + 1 < i1) ld(it + 1, (it + 1) & 1);
                        __builtin_amdgcn_sched_barrier(0);
                        const f32x4 av = fa[it & 1];
                        if (it < NG) {
                            const f32x4 bv = fb[it & 1];
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                if (i & 1) s1 = mfma16(av[i], bv[i], s1);
                                else s0 = mfma16(av[i], bv[i], s0);
                            }
                            if (it == NG - 1) {
#pragma unroll
                                for (int j = 0; j < REM; ++j) {
                                    if (j & 1) s1 = mfma16(rq[j], rk[j], s1);
                                    else s0 = mfma16(rq[j], rk[j], s0);
                                }
                            }
                        } else {
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                if (i & 1) d1 = mfma16(av[i], vT[4 * (it - NG) + i], d1);
                                else d0 = mfma16(av[i], vT[4 * (it - NG) + i], d0);
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            };
            using IC0 = std::integral_constant<int, 0>;
            using ICG = std::integral_constant<int, NG>;
            using ICI = std::integral_constant<int, NI>;
            if constexpr (!PAIR) {
                ld(0, 0);
#pragma unroll
                for (int it = 0; it < NI; ++it) {
                    if (it + 1 < NI) ld(it + 1, (it + 1) & 1);
                    __builtin_amdgcn_sched_barrier(0);
                    const f32x4 av = fa[it & 1];
                    if (it < NG) {
                        const f32x4 bv = fb[it & 1];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            if (i & 1) s1 = mfma16(av[i], bv[i], s1);
                            else s0 = mfma16(av[i], bv[i], s0);
                        }
                        if (it == NG - 1) {
#pragma unroll
                            for (int j = 0; j < REM; ++j) {
                                if (j & 1) s1 = mfma16(rq[j], rk[j], s1);
                                else s0 = mfma16(rq[j], rk[j], s0);
                            }
                        }
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            if (i & 1) d1 = mfma16(av[i], vT[4 * (it - NG) + i], d1);
                            else d0 = mfma16(av[i], vT[4 * (it - NG) + i], d0);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
                if (role_v) run_items(ICG{}, ICI{});
                else run_items(IC0{}, ICG{});
            }
            if (NG == 0 && do_k) {
#pragma unroll
                for (int j = 0; j < REM; ++j) {
                    if (j & 1) s1 = mfma16(rq[j], rk[j], s1);
                    else s0 = mfma16(rq[j], rk[j], s0);
                }
            }
        }
        if constexpr (!PAIR) {
            float pd[4], ds[4];
            {
                const f32x4 lse4 = *reinterpret_cast<const f32x4*>(ax + 4 * lq);
                const f32x4 del4 = *reinterpret_cast<const f32x4*>(ax + 16 + 4 * lq);
                const f32x4 key4 = *reinterpret_cast<const f32x4*>(ax + 32 + 4 * lq);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float p = tok_ok ? __expf(s0[r] + s1[r] - lse4[r]) : 0.f;
                    float dsc = 1.f;
                    if (DROP && (!DET || a.thresh) && !(GZ_ABL & 4))
                        dsc = (gz_mix32(__float_as_uint(key4[r]) ^ tokmul) >= a.thresh) ? a.scale : 0.f;
                    pd[r] = p * dsc;
                    ds[r] = p * ((d0[r] + d1[r]) * dsc - del4[r]);
                }
            }
            // publish this wave's dS block (label 4lq + r, token 16*wave + lr) into the step's dS tile
            {
                float* dw = dSs + cur * BWD_BL * RSD + (4 * lq) * RSD + (lr & 3) * KSD + 4 * wave + (lr >> 2);
#pragma unroll
                for (int r = 0; r < 4; ++r) dw[r * RSD] = ds[r];
            }
            // dV[tok, f] += sum_l Pd[l, tok] dO[l, f] ;  dK[tok, e] += sum_l dS[l, tok] Q[l, e].  k step r pairs labels
            // {4q + r}: A = pd[r] / ds[r] straight from the S-tile registers, B = row 4lq + r of the dO / Q image.
            if (!(GZ_ABL & 2)) {
                constexpr int FG4 = NFT / 4 * 4, EG4 = NET / 4 * 4;
                constexpr bool F2 = (NFT - FG4) >= 2, F1 = ((NFT - FG4) & 1) != 0;
                constexpr bool E2 = (NET - EG4) >= 2, E1 = ((NET - EG4) & 1) != 0;
                // B fragments of k step r: one float per accumulator tile, read one k step ahead of its MFMAs
                float bo[2][NFT], bq[2][NET];
                auto ldrow = [&](int r, int slot) {
                    const float* orow = to + (4 * lq + r) * RSO;
                    const float* qrow = tq + (4 * lq + r) * RSQ;
#pragma unroll
                    for (int jg = 0; jg < NFT / 4; ++jg) {
                        const f32x4 v4 = *reinterpret_cast<const f32x4*>(orow + 64 * jg + x4[r >> 1]);
#pragma unroll
                        for (int e = 0; e < 4; ++e) bo[slot][4 * jg + e] = v4[e];
                    }
                    if (F2) {
                        const f32x2 v2 = *reinterpret_cast<const f32x2*>(orow + 16 * FG4 + x2[r >> 1]);
                        bo[slot][FG4] = v2[0];
                        bo[slot][FG4 + 1] = v2[1];
                    }
                    if (F1) bo[slot][NFT - 1] = orow[16 * (FG4 + (F2 ? 2 : 0)) + x1[r >> 1]];
#pragma unroll
                    for (int jg = 0; jg < NET / 4; ++jg) {
                        const f32x4 v4 = *reinterpret_cast<const f32x4*>(qrow + 64 * jg + x4[r >> 1]);
#pragma unroll
                        for (int e = 0; e < 4; ++e) bq[slot][4 * jg + e] = v4[e];
                    }
                    if (E2) {
                        const f32x2 v2 = *reinterpret_cast<const f32x2*>(qrow + 16 * EG4 + x2[r >> 1]);
                        bq[slot][EG4] = v2[0];
                        bq[slot][EG4 + 1] = v2[1];
                    }
                    if (E1) bq[slot][NET - 1] = qrow[16 * (EG4 + (E2 ? 2 : 0)) + x1[r >> 1]];
                };
                ldrow(0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (r + 1 < 4) ldrow(r + 1, (r + 1) & 1);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < NFT; ++j) dV[j] = mfma16(pd[r], bo[r & 1][j], dV[j]);
#pragma unroll
                    for (int j = 0; j < NET; ++j) dK[j] = mfma16(ds[r], bq[r & 1][j], dK[j]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        } else {
            // ---- PAIR: the K-role wave owns P, the mask and dS; the V-role wave owns dP.  Exchange through LDS.
            float* xch_pd = sm + P::oXch + (wave * 64 + lane) * 4;             // written by the K-role wave
            float* xch_dp = sm + P::oXch + NW * 256 + (wave * 64 + lane) * 4;  // written by the V-role wave
            float pr[4], dsc[4];
            if (!role_v) {
                const f32x4 lse4 = *reinterpret_cast<const f32x4*>(ax + 4 * lq);
                const f32x4 key4 = *reinterpret_cast<const f32x4*>(ax + 32 + 4 * lq);
                f32x4 pdv;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    pr[r] = tok_ok ? __expf(s0[r] + s1[r] - lse4[r]) : 0.f;
                    dsc[r] = 1.f;
                    if (DROP && (!DET || a.thresh))
                        dsc[r] = (gz_mix32(__float_as_uint(key4[r]) ^ tokmul) >= a.thresh) ? a.scale : 0.f;
                    pdv[r] = pr[r] * dsc[r];
                }
                *reinterpret_cast<f32x4*>(xch_pd) = pdv;
            } else {
                *reinterpret_cast<f32x4*>(xch_dp) = d0 + d1;
            }
            if (!(GZ_ABL & 64)) __syncthreads();  // both tiles of every pair are in LDS
            constexpr int FG4 = NFT / 4 * 4, EG4 = NET / 4 * 4;
            constexpr bool F2 = (NFT - FG4) >= 2, F1 = ((NFT - FG4) & 1) != 0;
            constexpr bool E2 = (NET - EG4) >= 2, E1 = ((NET - EG4) & 1) != 0;
            float bb[2][NACC];  // B fragments of k step r, read one k step ahead of its MFMAs
            if (!role_v) {
                const f32x4 dp4 = *reinterpret_cast<const f32x4*>(xch_dp);
                const f32x4 del4 = *reinterpret_cast<const f32x4*>(ax + 16 + 4 * lq);
                float ds[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) ds[r] = pr[r] * (dp4[r] * dsc[r] - del4[r]);
                {
                    float* dw = dSs + cur * BWD_BL * RSD + (4 * lq) * RSD + (lr & 3) * KSD + 4 * wave + (lr >> 2);
#pragma unroll
                    for (int r = 0; r < 4; ++r) dw[r * RSD] = ds[r];
                }
                auto ldq = [&](int r, int slot) {
                    const float* qrow = tq + (4 * lq + r) * RSQ;
#pragma unroll
                    for (int jg = 0; jg < NET / 4; ++jg) {
                        const f32x4 v4 = *reinterpret_cast<const f32x4*>(qrow + 64 * jg + x4[r >> 1]);
#pragma unroll
                        for (int e = 0; e < 4; ++e) bb[slot][4 * jg + e] = v4[e];
                    }
                    if (E2) {
                        const f32x2 v2 = *reinterpret_cast<const f32x2*>(qrow + 16 * EG4 + x2[r >> 1]);
                        bb[slot][EG4] = v2[0];
                        bb[slot][EG4 + 1] = v2[1];
                    }
                    if (E1) bb[slot][NET - 1] = qrow[16 * (EG4 + (E2 ? 2 : 0)) + x1[r >> 1]];
                };
                ldq(0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (r + 1 < 4) ldq(r + 1, (r + 1) & 1);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < NET; ++j) acc[j] = mfma16(ds[r], bb[r & 1][j], acc[j]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
                const f32x4 pd4 = *reinterpret_cast<const f32x4*>(xch_pd);
                auto ldo = [&](int r, int slot) {
                    const float* orow = to + (4 * lq + r) * RSO;
#pragma unroll
                    for (int jg = 0; jg < NFT / 4; ++jg) {
                        const f32x4 v4 = *reinterpret_cast<const f32x4*>(orow + 64 * jg + x4[r >> 1]);
#pragma unroll
                        for (int e = 0; e < 4; ++e) bb[slot][4 * jg + e] = v4[e];
                    }
                    if (F2) {
                        const f32x2 v2 = *reinterpret_cast<const f32x2*>(orow + 16 * FG4 + x2[r >> 1]);
                        bb[slot][FG4] = v2[0];
                        bb[slot][FG4 + 1] = v2[1];
                    }
                    if (F1) bb[slot][NFT - 1] = orow[16 * (FG4 + (F2 ? 2 : 0)) + x1[r >> 1]];
                };
                ldo(0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (r + 1 < 4) ldo(r + 1, (r + 1) & 1);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < NFT; ++j) acc[j] = mfma16(pd4[r], bb[r & 1][j], acc[j]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        // dQ of the PREVIOUS step (its dS tile was completed before the last barrier)
        if (it > 0 && !(GZ_ABL & 1)) dq_tile(cur ^ 1);
        if (more && !(GZ_ABL & 8)) commit(cur ^ 1);  // next step's tiles: that buffer's readers passed the last barrier
        l0_p2 = l0_p1;
        l0_p1 = l0;
        __syncthreads();
    };
    for (int it = 0; it < nsteps; it += 2) {
        step_body(std::integral_constant<int, 0>{}, it);
        if (it + 1 < nsteps) step_body(std::integral_constant<int, 1>{}, it + 1);
    }
    if (nsteps > 0 && !(GZ_ABL & 1)) {  // drain: dQ of the last step, flushes of the last two
        const int last = (nsteps - 1) & 1;
        dq_tile(last);
        if (nsteps > 1) dq_flush(last ^ 1, l0_p2);
        __syncthreads();
        dq_flush(last, l0_p1);
    }
    __syncthreads();
    if (GZ_ABL & 2) dK[0][0] += 0.f;
    (void)dK, (void)dV, (void)acc;
    // epilogue: dK / dV tiles -> token rows through per-wave LDS staging, row-contiguous atomics
    if (!(GZ_ABL & 16)) {
        float* st = sm + wv * 16 * P::STW;
        // DET: label split sp owns slab sp; its token rows of note b are written by this workgroup alone
        float* gk = (DET ? a.dpx_slab + (long long)sp * a.B * a.Tp * a.E : a.dPX) + (long long)b * a.Tp * a.E;
        float* gv = (DET ? a.dx_slab + (long long)sp * a.B * a.Tp * a.F : a.dX) + (long long)b * a.Tp * a.F;
        if constexpr (!PAIR) {
            flush_rows<NET, P::STW, DET>(dK, st, lane, tok0, a.Tp, a.E, gk, a.E);
            flush_rows<NFT, P::STW, DET>(dV, st, lane, tok0, a.Tp, a.F, gv, a.F);
        } else {
            if (!role_v) flush_rows<NET, P::STW, DET>(acc, st, lane, tok0, a.Tp, a.E, gk, a.E);
            else flush_rows<NFT, P::STW, DET>(acc, st, lane, tok0, a.Tp, a.F, gv, a.F);
        }
    }
}

template <int ES, int NFT>
static int launch_fwd(const AttnFwdArgs& a, hipStream_t st) {
    size_t smem = (size_t)FwdLds<ES, NFT>::floats * 4;
    const int nLT = cdiv(a.L, 16 * FWD_WAVES);
    const dim3 grid(a.B * nLT), block(FWD_WAVES * 64);
    const bool vec = (a.E & 3) == 0 && (a.F & 3) == 0 &&
                     (((uintptr_t)a.PX | (uintptr_t)a.X | (uintptr_t)a.Q) & 15) == 0;
    if (vec && a.thresh) hipLaunchKernelGGL((label_attn_fwd_kernel<ES, NFT, 4, true>), grid, block, smem, st, a);
    else if (vec) hipLaunchKernelGGL((label_attn_fwd_kernel<ES, NFT, 4, false>), grid, block, smem, st, a);
    else if (a.thresh) hipLaunchKernelGGL((label_attn_fwd_kernel<ES, NFT, 1, true>), grid, block, smem, st, a);
    else hipLaunchKernelGGL((label_attn_fwd_kernel<ES, NFT, 1, false>), grid, block, smem, st, a);
    GZSL_CHECK_LAUNCH();
    return 0;
}

template <int ES, int NFT>
static int launch_bwd(const AttnBwdArgs& a, hipStream_t st) {
    // 8 waves (128 tokens) per workgroup while the keys fit LDS beside the streamed tiles; 4 waves for E > 100
    // (token groups of 16; with PAIR = wave pairs, the workgroup has 8 waves either way)
    constexpr int NW = (ES <= 25) ? 8 : 4;
    constexpr bool PAIR = ES > 25;
    using P = BwdLds<ES, NFT, NW, PAIR>;
    static_assert(P::floats * 4 <= 160 * 1024, "backward LDS plan exceeds 160 KiB");
    const int nTT = cdiv(a.Tp, 16 * NW);
    const dim3 grid(a.B * nTT * a.nsplit), block(P::NWV * 64);
    const bool vec = (a.E & 3) == 0 && (a.F & 3) == 0 &&
                     (((uintptr_t)a.Q | (uintptr_t)a.dO | (uintptr_t)a.PX | (uintptr_t)a.X) & 15) == 0;
    const size_t smem = (size_t)P::floats * 4;
#define GZ_BWD(G_, DROP_, DET_) \
    hipLaunchKernelGGL((label_attn_bwd_kernel<ES, NFT, NW, G_, PAIR, DROP_, DET_>), grid, block, smem, st, a)
    if (a.dq_slab) {  // deterministic mode: dropout stays a runtime test there (fewer instantiations; it is not the fast path)
        if (vec) GZ_BWD(4, true, true);
        else GZ_BWD(1, true, true);
    } else if (vec && a.thresh) GZ_BWD(4, true, false);
    else if (vec) GZ_BWD(4, false, false);
    else if (a.thresh) GZ_BWD(1, true, false);
    else GZ_BWD(1, false, false);
#undef GZ_BWD
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
    if ((long long)L * F >= (1ll << 31) || (long long)L * E >= (1ll << 31)) return GZSL_EINVAL;  // 32-bit in-note offsets
    AttnBwdArgs a{dO, PX, X, Q, LSE, DELTA, dPX, dX, dQ, B, Tp, E, F, L, l_offset, L_total, nsplit, seed, drop_thresh,
                  drop_scale, nullptr, nullptr, nullptr};
    hipStream_t st = (hipStream_t)stream;
    GZ_DISPATCH(launch_bwd, a, st);
}

static int bwd_tokens_per_wg(int E) { return E <= 100 ? 128 : 64; }  // launch_bwd's NW * 16

extern "C" long long gzsl_label_attn_bwd_det_work_floats(int B, int Tp, int E, int F, int L, int nsplit) {
    if (B <= 0 || Tp <= 0 || E <= 0 || F <= 0 || L <= 0 || nsplit <= 0) return 0;
    const long long nTT = cdiv(Tp, bwd_tokens_per_wg(E));
    return (long long)B * nTT * L * E + (long long)nsplit * B * Tp * (E + F);
}

static int launch_bwd_any(const AttnBwdArgs& a, hipStream_t st) { GZ_DISPATCH(launch_bwd, a, st); }

extern "C" int gzsl_label_attn_bwd_det(const float* dO, const float* PX, const float* X, const float* Q, const float* LSE,
                                       const float* DELTA, float* dPX, float* dX, float* dQ, int B, int Tp, int E, int F,
                                       int L, int l_offset, int L_total, int nsplit, uint32_t seed, uint32_t drop_thresh,
                                       float drop_scale, float* work, long long work_floats, void* stream) {
    if (B <= 0 || L <= 0) return 0;
    if (Tp <= 0 || E <= 0 || F <= 0 || nsplit <= 0 || !work) return GZSL_EINVAL;
    if ((long long)L * F >= (1ll << 31) || (long long)L * E >= (1ll << 31)) return GZSL_EINVAL;
    if (work_floats < gzsl_label_attn_bwd_det_work_floats(B, Tp, E, F, L, nsplit)) return GZSL_EINVAL;
    const long long nTT = cdiv(Tp, bwd_tokens_per_wg(E));
    float* dq_slab = work;
    float* dpx_slab = dq_slab + (long long)B * nTT * L * E;
    float* dx_slab = dpx_slab + (long long)nsplit * B * Tp * E;
    AttnBwdArgs a{dO, PX, X, Q, LSE, DELTA, dPX, dX, dQ, B, Tp, E, F, L, l_offset, L_total, nsplit, seed, drop_thresh,
                  drop_scale, dq_slab, dpx_slab, dx_slab};
    hipStream_t st = (hipStream_t)stream;
    int rc = launch_bwd_any(a, st);
    if (rc) return rc;
    // fixed-order sums of the slabs, ADDED to the caller's buffers (same contract as the atomic entry point)
    rc = gzsl_splitk_reduce(dq_slab, (long long)L * E, (int)(B * nTT), dQ, L, E, E, nullptr, GZSL_ACT_NONE, 1, 0, 0, stream);
    if (rc) return rc;
    rc = gzsl_splitk_reduce(dpx_slab, (long long)B * Tp * E, nsplit, dPX, B * Tp, E, E, nullptr, GZSL_ACT_NONE, 1, 0, 0, stream);
    if (rc) return rc;
    return gzsl_splitk_reduce(dx_slab, (long long)B * Tp * F, nsplit, dX, B * Tp, F, F, nullptr, GZSL_ACT_NONE, 1, 0, 0, stream);
}
