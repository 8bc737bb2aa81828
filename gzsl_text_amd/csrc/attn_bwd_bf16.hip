// Backward of the fused per-label attention with every fp32 product formed on the bf16 matrix pipe (bf16x3.h): the same
// contract as gzsl_label_attn_bwd (attn.hip; replaces autograd of icd_modules.py:137-142) -- inputs, outputs and
// accumulation stay fp32, each operand is split exactly into three bf16 planes and each product is six
// v_mfma_f32_16x16x32_bf16 (6/16 of the matrix time of v_mfma_f32_16x16x4_f32).  Opt-in on the host side
// (GZSL_ATTN_BF16X6=1); the headline benchmark stays on the exact-fp32 kernel.
//
// Structure (a new kernel, not a restatement of the fp32 one: k = 32 per MFMA reshapes every tile):
//   * a pre-pass splits Q and dO once per call into per-32-label TILE IMAGES that are byte for byte the LDS layout
//     (three bf16 planes, row-major, padded rows) and appends each tile's lse / delta / dropout row keys; the sweep's
//     staging is a flat 16-byte copy;
//   * workgroup = 64 tokens of one note = 4 token groups x a PAIR of waves, two waves per SIMD: the K-role wave of a
//     group holds the K planes (MFMA B fragments) and dK, computes S = Q.K^T, P, dS, dK and dQ; the V-role
//     wave holds the V planes and dV and computes dP = dO.V^T and dV.  (One wave owning all of a group's state needs
//     212 persistent + ~250 working registers: at one wave per SIMD hipcc parked the B fragments in AGPRs behind ~260
//     copies per step and nothing hid the LDS latency or the softmax arithmetic: 3.7 ms.)  The pair exchanges the
//     16 x 32 Pd planes and dP tile through LDS across one barrier per step;
//   * tokens sit on the MFMA lane axis, so the S / dP accumulators of two 16-label tiles ARE the k = 32 A fragments of
//     dV += Pd^T.dO and dK += dS^T.Q (k order: label = 16 (j >> 2) + 4 g + (j & 3) for element j of lane group g);
//   * ONE image per tile serves both ways of reading it: by rows (ds_read_b128: A operands of S and dP) and by columns
//     (ds_read_b64_tr_b16: B operands of the dV / dK products); row strides 288 B (Q) and 480 B (dO) make both
//     conflict-free without a swizzle (bank rule of MI355X_MICROARCH.md, checked by brute force);
//   * dS crosses LDS once, as bf16 planes [token][label] (two buffers); dQ += dS.K is the K-role waves' own product, one
//     16-column tile pair of e per wave: its B fragments (K^T, k = the workgroup's 64 tokens in the order 16 h + 4 g + q
//     that the transposed reads of the dS image deliver) never change over the sweep and live in registers.  Its first k
//     step runs in the step's last phase (beside the LDS write pass of the next dO tile, when no other MFMA is in flight),
//     its second one at the end of the next step's dK.  The result leaves as PLAIN streamed stores into one slab per
//     (note, token tile); a second launch adds the slabs in a fixed order (dQ is bitwise reproducible).  As float atomics
//     the 2.5 GB of partial tiles of a 64-token tiling sat at the chip's atomic rate (1.3 TB/s: +1.05 ms) and made every
//     workgroup start its sweep at another tile, which took the 13 GB of tile reads out of L2;
//   * the Q image has two LDS buffers filled by LDS-DMA (global_load_lds_dwordx4, issued at the top of a step for the next
//     one); the dO image (46 KB) has one and goes through registers: loads issued between the MFMAs of the step, written
//     between its last two barriers;
//   * the dropout mask is the V role's job (hash in the shadow of dP's MFMAs): dP leaves for the K role with mask and
//     scale on it, the planes of P arrive unmasked and get whole bf16 pieces zeroed (exact), and the scale 1/(1-p) goes
//     onto dV once at the end of the sweep.
// Measured at cfg2 (B16 x L9000 x T'2491, E100, F200, dropout .2): pre-pass 76 us + sweep 2.7 ms + dQ reduce 0.41 ms
// against 4.55 ms for the exact-fp32 kernel (profiles/r03_attn_bf16x6_*).  In-kernel section timers
// (tools/prof_bwd_phases.py, BB_PROF builds) put a step at ~7900 clocks against 300 MFMAs x 16 = 4800: ~1150 in the
// phase that writes the next dO tile (LDS writes run at 64 B/clk), the rest in the two compute phases, where the two
// waves of a SIMD do not overlap but take turns (the one the arbiter favours runs at its solo rate of ~19 clocks per
// MFMA, the other one mostly after it; raising either role's priority moves the wait, not the sum) and each phase ends
// in a serial tail (softmax of the last label tile; dS before the first dK MFMA).  What was tried on top and did not
// pay is listed in DESIGN.md (round 3).
#include <type_traits>
#include "common.h"
#include "bf16x3.h"
#include "rng.h"
#include "../../include/gzsl_hip.h"

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
#define GZ_LDS __attribute__((address_space(3)))

#ifndef BB_ABL
#define BB_ABL 0  // diagnostic builds only: bit 0 no dQ stores, bit 1 no dQ product, bit 2 no staging (tiles stay as loaded once)
#endif
// BB_PROF (diagnostic builds only, tools/prof_bwd_phases.py): waves 0, 3 and 4 of every workgroup sum the shader-clock time
// of each section of a step (scalar registers only) and write the sums behind the dQ slabs.
#ifdef BB_PROF
#define BB_T(k) { const long long t_ = clock64(); pacc[k] += t_ - tlast; tlast = t_; }
#define BB_T_BEGIN long long pacc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; long long tlast = clock64();
#define BB_T_END { if (c.lane == 0) { long long* d_ = reinterpret_cast<long long*>(a.slab + (long long)a.B * c.nTT * a.nLT * BwdBf::NET * 512) + ((long long)blockIdx.x * BB_WAVES + c.wv) * 10; for (int k_ = 0; k_ < 10; ++k_) d_[k_] = pacc[k_]; } }
#else
#define BB_T(k)
#define BB_T_BEGIN
#define BB_T_END
#endif
#ifndef BB_DEPTH_K
#define BB_DEPTH_K 1  // items of fragments in flight in the K role's loops
#endif
#ifndef BB_DEPTH_V
#define BB_DEPTH_V 3  // items of fragments in flight in the V role's loops
#endif
#define BB_TG 4                 // token groups of 16 per workgroup
#define BB_WAVES (2 * BB_TG)    // waves [0, 4): K role, [4, 8): V role
#define BB_NT (BB_WAVES * 64)
#define BB_TOK (16 * BB_TG)     // tokens per workgroup
#define BB_BL 32                // labels per step = k of one MFMA

struct BwdBf {
    static constexpr int EKS = 4, FKS = 7;   // k steps of S (E <= 128) and dP (F <= 224)
    static constexpr int NET = 7, NFT = 13;  // 16-wide column tiles of dK / dQ (E <= 112) and dV (F <= 208)
    static constexpr int RQ = 288, RO = 480; // row strides (bytes) of the Q / dO tile images
    static constexpr int PLQ = BB_BL * RQ, PLO = BB_BL * RO;
    static constexpr int QB = 3 * PLQ;                   // 27648 bytes of Q planes ...
    static constexpr int QCH = 1792, QBP = QCH * 16;     // ... padded to 1792 16-byte chunks
    static constexpr int OB = 3 * PLO;                   // 46080
    static constexpr int AUXB = 3 * BB_BL * 4;           // lse[32], delta[32], dropout row keys[32]
    static constexpr int TO = OB + AUXB, OCH = TO / 16;  // 46464 bytes = 2904 chunks
    static constexpr int RS = 72, PLS = BB_TOK * RS;     // dS image [64 tokens][32 labels] x 3 planes, two buffers
    static constexpr int XPD = BB_TG * 64 * 16;          // one plane of the Pd exchange (16 bytes per lane)
    static constexpr int oQ = 0, oO = 2 * QBP, oS = oO + TO, oXP = oS + 2 * 3 * PLS, oXD = oXP + 3 * XPD, total = oXD + 2 * XPD;
    // The Q image has TWO buffers and is filled by LDS-DMA (global_load_lds_dwordx4: 1 KB per wave instruction, QDMA of them
    // per K-role wave and step; no registers, no write pass).  The dO image has one buffer and goes through registers, one step
    // ahead: the 256 K-role threads take its first NSK x 256 16-byte chunks, the 256 V-role threads the remaining ones.
    static constexpr int QDMA = QBP / 1024 / BB_TG, NSK = 6, OCK = NSK * 256, NSV = (OCH - OCK + 255) / 256;
};
static_assert(BwdBf::QDMA * BB_TG * 1024 == BwdBf::QBP && BwdBf::NSV == 6, "staging plan");
static_assert(BwdBf::total <= 160 * 1024, "LDS plan");
static_assert(BwdBf::TO % 16 == 0 && BwdBf::oS % 16 == 0 && BwdBf::oXP % 16 == 0, "16-byte aligned regions");

__device__ __forceinline__ s16x4 lds_tr(const unsigned char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((GZ_LDS s16x4*)p);
}
__device__ __forceinline__ bf16x8 join8(s16x4 lo, s16x4 hi) {
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

// ---------------------------------------------------------------------------------------------------------- pre-pass
// grid (nLT, B + 1): y < B packs dO[y] tile + aux, y == B packs the Q tile.
__global__ __launch_bounds__(256) void attn_bwd_pack_bf16_kernel(const float* __restrict__ Q, const float* __restrict__ dO,
                                                                 const float* __restrict__ LSE, const float* __restrict__ DELTA,
                                                                 unsigned char* __restrict__ Qp, unsigned char* __restrict__ Op,
                                                                 int B, int E, int F, int L, int nLT, uint32_t seed,
                                                                 uint32_t l_offset, uint32_t L_total) {
    using T = BwdBf;
    const int lt = blockIdx.x, y = blockIdx.y, l0 = lt * BB_BL;
    if (y == B) {
        unsigned char* out = Qp + (long long)lt * T::QBP;
        for (int it = threadIdx.x; it < BB_BL * 16; it += blockDim.x) {
            const int r = it >> 4, c = it & 15, l = l0 + r;
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int e = 8 * c + j;
                v[j] = (l < L && e < E) ? Q[(long long)l * E + e] : 0.f;
            }
            bf16x8 pl[3];
            split3_x8(v, pl);
#pragma unroll
            for (int p = 0; p < 3; ++p) *reinterpret_cast<bf16x8*>(out + p * T::PLQ + r * T::RQ + 16 * c) = pl[p];
        }
        return;
    }
    unsigned char* out = Op + ((long long)y * nLT + lt) * T::TO;
    const float* dOb = dO + (long long)y * L * F;
    for (int it = threadIdx.x; it < BB_BL * 28; it += blockDim.x) {
        const int r = it / 28, c = it % 28, l = l0 + r;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int f = 8 * c + j;
            v[j] = (l < L && f < F) ? dOb[(long long)l * F + f] : 0.f;
        }
        bf16x8 pl[3];
        split3_x8(v, pl);
#pragma unroll
        for (int p = 0; p < 3; ++p) *reinterpret_cast<bf16x8*>(out + p * T::PLO + r * T::RO + 16 * c) = pl[p];
    }
    if (threadIdx.x < BB_BL) {
        const int l = l0 + threadIdx.x;
        float* aux = reinterpret_cast<float*>(out + T::OB);
        const bool ok = l < L;
        const long long o = (long long)y * L + (ok ? l : L - 1);
        aux[threadIdx.x] = ok ? LSE[o] : INFINITY;  // +inf -> P = 0 for labels past L
        aux[32 + threadIdx.x] = ok ? DELTA[o] : 0.f;
        aux[64 + threadIdx.x] = __uint_as_float(gz_row_key(seed, GZSL_STREAM_ATTN, (uint32_t)y * L_total + l_offset + (uint32_t)l));
    }
}

struct AttnBwdBfArgs {
    const unsigned char* Qp;  // [nLT][QBP]
    const unsigned char* Op;  // [B][nLT][TO]
    const float* PX;
    const float* X;
    float* dPX;
    float* dX;
    float* dQ;
    float* slab;  // [B * nTT][nLT][NET][2 u][4 r][64 lanes] partial dQ tiles, plain stores; summed by attn_bwd_dq_reduce_kernel
    int B, Tp, E, F, L, nLT, nsplit;
    uint32_t thresh;
    float scale;
};

// eight consecutive floats row[c0 .. c0+7] (zero past `width` or when !ok); vec: width % 4 == 0 and 16-byte aligned rows
__device__ __forceinline__ void load8(const float* row, int c0, int width, bool ok, bool vec, float (&v)[8]) {
    if (vec) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int c = c0 + 4 * h;
            const f32x4 t = *reinterpret_cast<const f32x4*>(row + min(c, width - 4));
            const float keep = (ok && c < width) ? 1.f : 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[4 * h + j] = t[j] * keep;
        }
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = c0 + j;
            v[j] = row[min(c, width - 1)] * ((ok && c < width) ? 1.f : 0.f);
        }
    }
}

// What both roles of a workgroup share: its coordinates and the lane constants of the LDS read patterns.
struct BbCtx {
    unsigned char* sm;
    int tid, lane, wv, tg, n, g, trq, trp;
    int b, tt, nTT, tok0, tok, step_beg, nsteps;
    bool tok_ok;
};

// epilogue of either role: accumulator tiles (column 16j + n, token rows 4g + r) -> float atomics
template <int NTILE>
__device__ __forceinline__ void bb_flush(const f32x4 (&acc)[NTILE], float* gp, int W, const BbCtx& c, int Tp) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int t = 4 * c.g + r;
        if (c.tok0 + t < Tp) {
#pragma unroll
            for (int j = 0; j < NTILE; ++j)
                if (16 * j + c.n < W) atomicAdd(gp + (long long)t * W + 16 * j + c.n, acc[j][r]);
        }
    }
}

// ---- K role: K planes (B fragments of S; element j of k step ks <-> e = 32 ks + 8 g + j), S, P, dS, dK, dQ.  Brings in
// the Q tile (LDS-DMA) and the first OCK chunks of the dO tile.  (Dropout does not appear here: the mask lives in the V role.)
__device__ __forceinline__ void bb_role_k(const AttnBwdBfArgs& a, const BbCtx& c) {
    using T = BwdBf;
    constexpr int EKS = T::EKS, NET = T::NET;
    unsigned char* sm = c.sm;
    const int n = c.n, g = c.g;
    bf16x8 breg[EKS][3];
    f32x4 acc[NET];
    {
        const bool vec = (a.E & 3) == 0 && (((uintptr_t)a.PX) & 15) == 0;
        const float* krow = a.PX + ((long long)c.b * a.Tp + min(c.tok, a.Tp - 1)) * a.E;
#pragma unroll
        for (int ks = 0; ks < EKS; ++ks) {
            float v[8];
            load8(krow, 32 * ks + 8 * g, a.E, c.tok_ok, vec, v);
            split3_x8(v, breg[ks]);
        }
    }
    // dQ[l, e] += sum_tok dS[l, tok] K[tok, e] (k = the workgroup's 64 tokens): K-role wave tg owns the column tiles
    // e in [16 tg, 16 tg + 16) and [16 (tg + 4), ..) of both label tiles.  Its B fragments never change over the sweep:
    // element j of k step ks <-> token 32 ks + 16 (j >> 2) + 4 g + (j & 3) (the order the transposed reads of the dS
    // image deliver), column e = 16 jt + n -- held in registers, read once from the fp32 keys.
    bf16x8 kq[2][BB_TOK / 32][3];
    {
        const float* kb = a.PX + ((long long)c.b * a.Tp + (long long)c.tt * BB_TOK) * a.E;
        const int trows = a.Tp - c.tt * BB_TOK;  // valid token rows of this tile
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int ks = 0; ks < BB_TOK / 32; ++ks) {
                const int e = 16 * (c.tg + 4 * jj) + n;
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int t = 32 * ks + 16 * (j >> 2) + 4 * g + (j & 3);
                    v[j] = kb[(long long)min(t, trows - 1) * a.E + min(e, a.E - 1)] * ((t < trows && e < a.E) ? 1.f : 0.f);
                }
                split3_x8(v, kq[jj][ks]);
            }
    }
    const bool two = c.tg + 4 < NET;  // wave-uniform
    f32x4 q2[2][2];                   // [column tile][label tile]
    const unsigned char* str = sm + T::oS + (4 * g + c.trq) * T::RS + 8 * c.trp;
    // One k step (32 tokens) of the product for the step whose dS image is buffer `buf`.  Both label tiles' A fragments
    // (dS image, read transposed) are fetched up front by dq_fetch and the six terms go round the accumulators, so neither
    // an LDS round trip nor a dependent MFMA sits between two MFMAs (each accumulator still sees the order of mfma_x6).
    bf16x8 dqa[2][3];
    auto dq_fetch = [&](int buf, int ks) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int p = 0; p < 3; ++p)
                dqa[u][p] = join8(lds_tr(str + (3 * buf + p) * T::PLS + (32 * ks) * T::RS + 32 * u),
                                  lds_tr(str + (3 * buf + p) * T::PLS + (32 * ks + 16) * T::RS + 32 * u));
    };
    auto dq_mfma = [&](int ks) {
        constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};  // the term order of mfma_x6
        if (two) {
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    q2[0][u] = mfma_bf16(dqa[u][PA[t]], kq[0][ks][PB[t]], q2[0][u]);
                    q2[1][u] = mfma_bf16(dqa[u][PA[t]], kq[1][ks][PB[t]], q2[1][u]);
                }
        } else {
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
                for (int u = 0; u < 2; ++u) q2[0][u] = mfma_bf16(dqa[u][PA[t]], kq[0][ks][PB[t]], q2[0][u]);
        }
    };
    // The tiles (column e = 16 jt + n, rows 16u + 4g + r) leave as PLAIN streamed stores into this (note, token tile)'s slab,
    // 256 contiguous bytes per instruction; attn_bwd_dq_reduce_kernel adds the slabs.
    auto dq_store = [&](int lt) {
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            if (jj == 1 && !two) continue;
            float* sl = a.slab + ((((long long)(c.b * c.nTT + c.tt) * a.nLT + lt) * NET + c.tg + 4 * jj) * 8) * 64 + c.lane;
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (!(BB_ABL & 1)) __builtin_nontemporal_store(q2[jj][u][r], sl + (4 * u + r) * 64);
        }
    };
    auto dq_zero = [&]() {
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) q2[jj][0] = q2[jj][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
#pragma unroll
    for (int j = 0; j < NET; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float tokmask = c.tok_ok ? 1.f : 0.f;

    u32x4 stg[T::NSK];
    const unsigned char* Ob = a.Op + (long long)c.b * a.nLT * T::TO;
    const int st_t = c.tid;  // K-role threads are 0..255
    auto issue1 = [&](int lt, int i) { stg[i] = reinterpret_cast<const u32x4*>(Ob + (long long)lt * T::TO)[st_t + 256 * i]; };
    auto commit = [&]() {
#pragma unroll
        for (int i = 0; i < T::NSK; ++i) reinterpret_cast<u32x4*>(sm + T::oO)[st_t + 256 * i] = stg[i];
    };
    // Q tile lt -> buffer buf: this wave's QDMA KB of the image, lane l of instruction i supplies bytes [16 l, 16 l + 16) of its KB
    auto issue_q = [&](int lt, int buf) {
        const unsigned char* src = a.Qp + (long long)lt * T::QBP + (c.tg * T::QDMA) * 1024 + c.lane * 16;
        unsigned char* dst = sm + T::oQ + buf * T::QBP + (c.tg * T::QDMA) * 1024;
#pragma unroll
        for (int i = 0; i < T::QDMA; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 1024 * i),
                                             (GZ_LDS void*)(dst + 1024 * i), 16, 0, 0);
    };
    const unsigned char* qrow0 = sm + T::oQ + n * T::RQ + 16 * g;                    // row read: row 16u + n, chunk 4ks + g
    const unsigned char* qtr0 = sm + T::oQ + (4 * g + c.trq) * T::RQ + 8 * c.trp;    // transposed: rows 16u + 4g + q, columns 16j + 4p ..
    const float* aux = reinterpret_cast<const float*>(sm + T::oO + T::OB);
    unsigned char* xpd = sm + T::oXP + (c.tg * 64 + c.lane) * 16;   // Pd planes: K role -> V role (same lane layout)
    const unsigned char* xdp = sm + T::oXD + (c.tg * 64 + c.lane) * 16;   // dP tiles:  V role -> K role

    const int nsteps = c.nsteps;
    if (nsteps > 0) {
        issue_q(c.step_beg, 0);
#pragma unroll
        for (int i = 0; i < T::NSK; ++i) issue1(c.step_beg, i);
        commit();
    }
    __syncthreads();
    BB_T_BEGIN
    for (int it = 0; it < nsteps; ++it) {
        const unsigned char* qrow = qrow0 + (it & 1) * T::QBP;
        const unsigned char* qtr = qtr0 + (it & 1) * T::QBP;
        // the next Q tile starts its way into the other buffer (last read before barrier B of step it - 1); the DMA is
        // drained by the wait hipcc puts in front of barrier X while an LDS-DMA is pending, ~3000 clocks from here
        // (no branch on "is there a next step": the last step stages a tile nobody reads, and the waits stay unconditional)
        const int lt_next = min(c.step_beg + it + 1, a.nLT - 1);
        if (!(BB_ABL & 4)) issue_q(lt_next, (it + 1) & 1);
        // ================= phase 1: S = Q.K^T for two 16-label tiles; A fragments read one item ahead of their MFMAs
        f32x4 hi[2], mid[2], lo[2];  // three accumulators per tile, by weight class of the cross terms (bf16x3.h)
#pragma unroll
        for (int u = 0; u < 2; ++u) hi[u] = mid[u] = lo[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        float pr[8];
        u32x4 pdh, pdm, pdl;  // the planes of P, built one label tile (two packed pairs) at a time
        auto softmax_tile = [&](int u) {
            const f32x4 lse4 = *reinterpret_cast<const f32x4*>(aux + 16 * u + 4 * g);
            const f32x4 s4 = hi[u] + (mid[u] + lo[u]);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                // (S accumulates from 0, lse subtracted here: with |lse| ~ 10 in the accumulator every partial sum would
                // round at ulp(10) and the error of P is amplified by the cancellation in dS = P (dP - delta))
                pr[4 * u + r] = __expf(s4[r] - lse4[r]) * tokmask;
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                unsigned h, m, l;
                split3_pk(pr[4 * u + 2 * i], pr[4 * u + 2 * i + 1], h, m, l);
                pdh[2 * u + i] = h, pdm[2 * u + i] = m, pdl[2 * u + i] = l;
            }
        };
        constexpr int DK = BB_DEPTH_K;  // items of fragments in flight (see the V role)
        bf16x8 fb[DK + 1][3];
        auto ldb1 = [&](int item, int p) {
            fb[item % (DK + 1)][p] = join8(lds_tr(qtr + p * T::PLQ + 32 * item), lds_tr(qtr + p * T::PLQ + 16 * T::RQ + 32 * item));
        };
        {
            bf16x8 fr[DK + 1][3];
            auto lda1 = [&](int item, int p) {
                fr[item % (DK + 1)][p] = *reinterpret_cast<const bf16x8*>(qrow + p * T::PLQ + 16 * (item / EKS) * T::RQ + 64 * (item % EKS));
            };
#pragma unroll
            for (int i = 0; i < DK; ++i)
#pragma unroll
                for (int p = 0; p < 3; ++p) lda1(i, p);
#pragma unroll
            for (int item = 0; item < 2 * EKS; ++item) {
                // (one register-staged chunk of the next dO tile per item: the loads' issue hides behind the MFMAs)
                if (item < T::NSK && !(BB_ABL & 4)) issue1(lt_next, item);
                __builtin_amdgcn_sched_barrier(0);
                mfma_x6_3acc_il(fr[item % (DK + 1)], breg[item % EKS], hi[item / EKS], mid[item / EKS], lo[item / EKS],
                                [&](int p) { if (item + DK < 2 * EKS) lda1(item + DK, p); });
                // P of the first label tile while the second tile's MFMAs run (two items after its last MFMA was issued)
                if (item == EKS + 1) softmax_tile(0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        BB_T(0)
        // the first dK fragments are on their way before the barrier: the Q tile does not change over it
#pragma unroll
        for (int i = 0; i < DK; ++i)
#pragma unroll
            for (int p = 0; p < 3; ++p) ldb1(i, p);
        softmax_tile(1);
        *reinterpret_cast<u32x4*>(xpd) = pdh;   // the planes of P (before dropout) go to the V-role wave of the pair
        *reinterpret_cast<u32x4*>(xpd + T::XPD) = pdm;
        *reinterpret_cast<u32x4*>(xpd + 2 * T::XPD) = pdl;
        BB_T(1)
        __syncthreads();  // X: both halves of every pair's exchange are in LDS
        BB_T(3)
        // ================= phase 2: dS and dK[tok, e] += sum_l dS[l, tok] Q[l, e]; B fragments by transposed reads
        {
            float ds[8];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const f32x4 dp4 = *reinterpret_cast<const f32x4*>(xdp + u * T::XPD);  // dP with the dropout mask and scale on it
                const f32x4 del4 = *reinterpret_cast<const f32x4*>(aux + 32 + 16 * u + 4 * g);
#pragma unroll
                for (int r = 0; r < 4; ++r) ds[4 * u + r] = pr[4 * u + r] * (dp4[r] - del4[r]);
            }
            bf16x8 dsp[3];
            split3_x8(ds, dsp);
            // dS planes -> [token][label] image `it & 1` (labels 16u + 4g .. +3 are one 8-byte store)
            unsigned char* sw = sm + T::oS + (it & 1) * 3 * T::PLS + (c.tg * 16 + n) * T::RS + 8 * g;
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                const u32x4 w = __builtin_bit_cast(u32x4, dsp[p]);
                *reinterpret_cast<u32x2*>(sw + p * T::PLS) = u32x2{w[0], w[1]};
                *reinterpret_cast<u32x2*>(sw + p * T::PLS + 32) = u32x2{w[2], w[3]};
            }
#pragma unroll
            for (int item = 0; item < NET; ++item) {
                // (the fragments of the previous step's second dQ k step ride along with the last item's)
                if (item == NET - 1 && it > 0 && !(BB_ABL & 2)) dq_fetch((it - 1) & 1, 1);
                __builtin_amdgcn_sched_barrier(0);
                acc[item] = mfma_x6_il(dsp, fb[item % (DK + 1)], acc[item], [&](int p) { if (item + DK < NET) ldb1(item + DK, p); });
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        BB_T(4)
        // dQ of the PREVIOUS step, second k step (dV is 78 MFMAs against dK's 42), then its tiles leave
        if (it > 0 && !(BB_ABL & 2)) {
            dq_mfma(1);
            dq_store(c.step_beg + it - 1);
        }
        BB_T(5)
        __syncthreads();  // B: every wave is done with tile `it`; this step's dS image is complete
        BB_T(6)
        // While the next dO tile goes from the staging registers into LDS (all eight waves: that pass runs at the LDS write
        // rate, ~730 clocks), the K-role waves run the first k step of this step's dQ product: the only MFMAs of the phase.
        if (!(BB_ABL & 2)) dq_fetch(it & 1, 0);
        if (!(BB_ABL & 4)) commit();
        if (!(BB_ABL & 2)) {
            dq_zero();
            dq_mfma(0);
        }
        BB_T(7)
        __syncthreads();  // A: dO tile it+1 is in LDS; the exchange buffers and the other dS image may be rewritten
        BB_T(8)
    }
    if (nsteps > 0 && !(BB_ABL & 2)) {
        dq_fetch((nsteps - 1) & 1, 1);
        dq_mfma(1);
        dq_store(c.step_beg + nsteps - 1);
    }
    BB_T_END
    bb_flush<NET>(acc, a.dPX + ((long long)c.b * a.Tp + c.tok0) * a.E, a.E, c, a.Tp);
}

// ---- V role: V planes (B fragments of dP), dP = dO.V^T, the dropout mask, dV[tok, f] += sum_l Pd[l, tok] dO[l, f].
// Stages the rest of the dO tile.
template <bool DROP>
__device__ __forceinline__ void bb_role_v(const AttnBwdBfArgs& a, const BbCtx& c) {
    using T = BwdBf;
    constexpr int FKS = T::FKS, NFT = T::NFT;
    unsigned char* sm = c.sm;
    const int n = c.n, g = c.g;
    bf16x8 breg[FKS][3];
    f32x4 acc[NFT];
    {
        const bool vec = (a.F & 3) == 0 && (((uintptr_t)a.X) & 15) == 0;
        const float* vrow = a.X + ((long long)c.b * a.Tp + min(c.tok, a.Tp - 1)) * a.F;
#pragma unroll
        for (int ks = 0; ks < FKS; ++ks) {
            float v[8];
            load8(vrow, 32 * ks + 8 * g, a.F, c.tok_ok, vec, v);
            split3_x8(v, breg[ks]);
        }
    }
#pragma unroll
    for (int j = 0; j < NFT; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};

    u32x4 stg[T::NSV];
    const unsigned char* Ob = a.Op + (long long)c.b * a.nLT * T::TO;
    const int st_t = c.tid - 256;  // V-role threads are 256..511
    auto issue1 = [&](int lt, int i) {
        stg[i] = reinterpret_cast<const u32x4*>(Ob + (long long)lt * T::TO)[min(T::OCK + st_t + 256 * i, T::OCH - 1)];
    };
    auto commit = [&]() {
#pragma unroll
        for (int i = 0; i < T::NSV; ++i)
            if (T::OCK + st_t + 256 * i < T::OCH) reinterpret_cast<u32x4*>(sm + T::oO)[T::OCK + st_t + 256 * i] = stg[i];
    };
    const unsigned char* orow = sm + T::oO + n * T::RO + 16 * g;
    const unsigned char* otr = sm + T::oO + (4 * g + c.trq) * T::RO + 8 * c.trp;
    const unsigned char* xpd = sm + T::oXP + (c.tg * 64 + c.lane) * 16;
    unsigned char* xdp = sm + T::oXD + (c.tg * 64 + c.lane) * 16;
    // The dropout mask is this role's job (its loops wait on LDS, the K role's on its own arithmetic): element (u, r) of a
    // lane is (label 16u + 4g + r, token n), kept iff gz_u32(row key, token) >= thresh.  dP leaves with mask and scale on
    // it; the planes of P are masked (whole bf16 pieces zeroed: exact) and the scale goes onto dV once, at the end.
    const float* aux = reinterpret_cast<const float*>(sm + T::oO + T::OB);
    const uint32_t tokmul = (uint32_t)c.tok * 0x85EBCA6Bu;  // gz_u32(key, tok) = mix32(key ^ tokmul)

    const int nsteps = c.nsteps;
    if (nsteps > 0) {
#pragma unroll
        for (int i = 0; i < T::NSV; ++i) issue1(c.step_beg, i);
        commit();
    }
    __syncthreads();
    BB_T_BEGIN
    for (int it = 0; it < nsteps; ++it) {
        const int lt_next = min(c.step_beg + it + 1, a.nLT - 1);
        u32x4 keep;  // per packed pair of P pieces: 0xFFFF in the half of a kept element
        // Fragment rings: an LDS read comes back ~250 clocks after its issue when all eight waves stream, an item's six
        // MFMAs take 96; with the fragments of only the next item in flight every item waited for its reads (the phase ran at
        // the SUM of its LDS and MFMA times).  DV items ahead, each ring DV + 1 sets of three planes.
        constexpr int DV = BB_DEPTH_V;
        bf16x8 fb[DV + 1][3];
        auto ldb1 = [&](int item, int p) {
            fb[item % (DV + 1)][p] = join8(lds_tr(otr + p * T::PLO + 32 * item), lds_tr(otr + p * T::PLO + 16 * T::RO + 32 * item));
        };
        auto ldb = [&](int item) {
#pragma unroll
            for (int p = 0; p < 3; ++p) ldb1(item, p);
        };
        {   // ================= phase 1: dP = dO.V^T (three accumulators per tile, by weight class as for S)
            f32x4 hi[2], mid[2], lo[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) hi[u] = mid[u] = lo[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            bf16x8 fr[DV + 1][3];
            auto lda1 = [&](int item, int p) {
                fr[item % (DV + 1)][p] = *reinterpret_cast<const bf16x8*>(orow + p * T::PLO + 16 * (item / FKS) * T::RO + 64 * (item % FKS));
            };
            auto lda = [&](int item) {
#pragma unroll
                for (int p = 0; p < 3; ++p) lda1(item, p);
            };
#pragma unroll
            for (int i = 0; i < DV; ++i) lda(i);
            f32x4 key4[2];
            if (DROP) {
                key4[0] = *reinterpret_cast<const f32x4*>(aux + 64 + 4 * g);
                key4[1] = *reinterpret_cast<const f32x4*>(aux + 64 + 16 + 4 * g);
            }
            float dsc[8];
#pragma unroll
            for (int item = 0; item < 2 * FKS; ++item) {
                __builtin_amdgcn_sched_barrier(0);
                mfma_x6_3acc_il(fr[item % (DV + 1)], breg[item % FKS], hi[item / FKS], mid[item / FKS], lo[item / FKS],
                                [&](int p) { if (item + DV < 2 * FKS) lda1(item + DV, p); });
                // one dropout hash per item, in the shadow of its MFMAs
                if (item < 8) {
                    dsc[item] = 1.f;
                    if (DROP) dsc[item] = (gz_mix32(__float_as_uint(key4[item >> 2][item & 3]) ^ tokmul) >= a.thresh) ? a.scale : 0.f;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // the first dV fragments are on their way before the barrier: the dO tile does not change over it
#pragma unroll
            for (int i = 0; i < DV; ++i) ldb(i);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                f32x4 d = hi[u] + (mid[u] + lo[u]);
#pragma unroll
                for (int r = 0; r < 4; ++r) d[r] *= dsc[4 * u + r];
                *reinterpret_cast<f32x4*>(xdp + u * T::XPD) = d;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
                keep[i] = DROP ? ((dsc[2 * i] != 0.f ? 0xFFFFu : 0u) | (dsc[2 * i + 1] != 0.f ? 0xFFFF0000u : 0u)) : 0xFFFFFFFFu;
        }
        BB_T(0)
        __syncthreads();  // X
        BB_T(3)
        {   // ================= phase 2: dV (of the masked P; the dropout scale is applied to the sums at the end)
            bf16x8 pdp[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                u32x4 w = *reinterpret_cast<const u32x4*>(xpd + p * T::XPD);
                if (DROP) w &= keep;
                pdp[p] = __builtin_bit_cast(bf16x8, w);
            }
#pragma unroll
            for (int item = 0; item < NFT; ++item) {
                // (one register-staged chunk of the next dO tile per item; issued in this phase, not in the first one, where
                // the registers are needed for the dP ring)
                if (item < T::NSV && !(BB_ABL & 4)) issue1(lt_next, item);
                __builtin_amdgcn_sched_barrier(0);
                acc[item] = mfma_x6_il(pdp, fb[item % (DV + 1)], acc[item], [&](int p) { if (item + DV < NFT) ldb1(item + DV, p); });
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        BB_T(4)
        __syncthreads();  // B
        BB_T(6)
        if (!(BB_ABL & 4)) commit();
        BB_T(7)
        __syncthreads();  // A
        BB_T(8)
    }
    BB_T_END
    if (DROP) {
#pragma unroll
        for (int j = 0; j < NFT; ++j) acc[j] *= a.scale;
    }
    bb_flush<NFT>(acc, a.dX + ((long long)c.b * a.Tp + c.tok0) * a.F, a.F, c, a.Tp);
}

template <bool DROP>
__global__ __launch_bounds__(BB_NT) void label_attn_bwd_bf16_kernel(AttnBwdBfArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    BbCtx c;
    c.sm = sm;
    c.tid = threadIdx.x;
    c.lane = c.tid & 63;
    c.wv = __builtin_amdgcn_readfirstlane(c.tid >> 6);  // scalar: branches on it keep EXEC all ones (transposed reads)
    c.tg = c.wv & (BB_TG - 1);                          // token group of this wave
    c.n = c.lane & 15;
    c.g = c.lane >> 4;
    c.trq = (c.lane >> 2) & 3;  // transposed read: lane 4q + p of a 16-lane group supplies row q, columns 4p..4p+3
    c.trp = c.lane & 3;
    c.nTT = (a.Tp + BB_TOK - 1) / BB_TOK;
    int sp;
    {
        // Workgroups are dealt to the 8 XCDs round-robin by id.  Those of one XCD take consecutive token tiles of the SAME
        // (note, label split) and sweep the label tiles in the same order at about the same time, so a tile image is
        // fetched into that XCD's L2 once and read from there by the others (13 GB of tile reads per launch at cfg2).
        const int id = blockIdx.x;
        if ((a.B & 7) == 0) {
            const int xcd = id & 7;
            int slot = id >> 3;
            c.tt = slot % c.nTT;
            slot /= c.nTT;
            sp = slot % a.nsplit;
            c.b = (slot / a.nsplit) * 8 + xcd;
        } else {
            c.tt = id % c.nTT;
            sp = (id / c.nTT) % a.nsplit;
            c.b = id / (c.nTT * a.nsplit);
        }
    }
    c.tok0 = c.tt * BB_TOK + c.tg * 16;
    c.tok = c.tok0 + c.n;
    c.tok_ok = c.tok < a.Tp;
    const int steps_per = (a.nLT + a.nsplit - 1) / a.nsplit;
    c.step_beg = sp * steps_per;
    c.nsteps = max(0, min(a.nLT, c.step_beg + steps_per) - c.step_beg);
    // the two roles run the same number of barriers per step (s_barrier counts arrivals, not program locations)
    if (c.wv >= BB_TG) bb_role_v<DROP>(a, c);
    else bb_role_k(a, c);
}

// dQ[l, e] += sum over the (note, token tile) slabs of the partial tiles label_attn_bwd_bf16_kernel stored, in slab order
// (fixed: this part of the result is bitwise reproducible).  One thread per tile element, 256-byte rows per wave.
__global__ __launch_bounds__(256) void attn_bwd_dq_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dQ,
                                                                 int nslab, long long per_slab, int E, int L) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= per_slab) return;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    const float* p = slab + i;
    int k = 0;
    for (; k + 4 <= nslab; k += 4) {
        s0 += p[(long long)k * per_slab];
        s1 += p[(long long)(k + 1) * per_slab];
        s2 += p[(long long)(k + 2) * per_slab];
        s3 += p[(long long)(k + 3) * per_slab];
    }
    for (; k < nslab; ++k) s0 += p[(long long)k * per_slab];
    const int lane = (int)(i & 63), ur = (int)((i >> 6) & 7), j = (int)((i >> 9) % BwdBf::NET);
    const long long lt = (i >> 9) / BwdBf::NET;
    const int l = (int)lt * BB_BL + 16 * (ur >> 2) + 4 * (lane >> 4) + (ur & 3), e = 16 * j + (lane & 15);
    if (l < L && e < E) dQ[(long long)l * E + e] += (s0 + s1) + (s2 + s3);
}

static bool bwd_bf_supported(int E, int F) { return E > 0 && F > 0 && E <= 16 * BwdBf::NET && F <= 16 * BwdBf::NFT; }

static long long bwd_bf_tile_bytes(int B, int L) {
    const long long nLT = cdiv(L, BB_BL);
    return nLT * BwdBf::QBP + (long long)B * nLT * BwdBf::TO;
}
// + the dQ slabs: one [nLT][7][512] float block per (note, 64-token tile)
static long long bwd_bf_slab_floats(int B, int Tp, int L) { return (long long)B * cdiv(Tp, BB_TOK) * cdiv(L, BB_BL) * BwdBf::NET * 512; }

extern "C" long long gzsl_label_attn_bwd_bf16x6_work_bytes(int B, int Tp, int E, int F, int L) {
    if (B <= 0 || L <= 0 || Tp <= 0 || !bwd_bf_supported(E, F)) return 0;
    return bwd_bf_tile_bytes(B, L) + 4 * bwd_bf_slab_floats(B, Tp, L);
}

extern "C" int gzsl_label_attn_bwd_bf16x6_dq_reduce(const void* work, float* dQ, int B, int Tp, int E, int F, int L, void* stream) {
    if (B <= 0 || L <= 0) return 0;
    if (Tp <= 0 || !bwd_bf_supported(E, F) || !work) return GZSL_EINVAL;
    const int nLT = cdiv(L, BB_BL), nTT = cdiv(Tp, BB_TOK);
    const float* slab = (const float*)((const unsigned char*)work + bwd_bf_tile_bytes(B, L));
    const long long per_slab = (long long)nLT * BwdBf::NET * 512;
    hipLaunchKernelGGL(attn_bwd_dq_reduce_kernel, dim3((unsigned)((per_slab + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       slab, dQ, B * nTT, per_slab, E, L);
    GZSL_CHECK_LAUNCH();
    return 0;
}

extern "C" int gzsl_label_attn_bwd_bf16x6(const float* dO, const float* PX, const float* X, const float* Q, const float* LSE,
                                          const float* DELTA, float* dPX, float* dX, float* dQ, int B, int Tp, int E, int F,
                                          int L, int l_offset, int L_total, int nsplit, uint32_t seed, uint32_t drop_thresh,
                                          float drop_scale, void* work, long long work_bytes, int defer_dq_reduce,
                                          void* stream) {
    if (B <= 0 || L <= 0) return 0;
    if (Tp <= 0 || nsplit <= 0 || !bwd_bf_supported(E, F) || !work || (((uintptr_t)work) & 15) != 0) return GZSL_EINVAL;
    const long long need = gzsl_label_attn_bwd_bf16x6_work_bytes(B, Tp, E, F, L);
    if (work_bytes < need) return GZSL_EINVAL;
    using T = BwdBf;
    const int nLT = cdiv(L, BB_BL), nTT = cdiv(Tp, BB_TOK);
    if (nsplit > nLT) nsplit = nLT;
    unsigned char* Qp = (unsigned char*)work;
    unsigned char* Op = Qp + (long long)nLT * T::QBP;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(attn_bwd_pack_bf16_kernel, dim3(nLT, B + 1), dim3(256), 0, st, Q, dO, LSE, DELTA, Qp, Op, B, E, F, L,
                       nLT, seed, (uint32_t)l_offset, (uint32_t)L_total);
    GZSL_CHECK_LAUNCH();
    float* slab = (float*)((unsigned char*)work + bwd_bf_tile_bytes(B, L));  // (tile bytes are a multiple of 16)
    AttnBwdBfArgs a{Qp, Op, PX, X, dPX, dX, dQ, slab, B, Tp, E, F, L, nLT, nsplit, drop_thresh, drop_scale};
    const dim3 grid(B * nTT * nsplit), block(BB_NT);
    if (drop_thresh) hipLaunchKernelGGL((label_attn_bwd_bf16_kernel<true>), grid, block, T::total, st, a);
    else hipLaunchKernelGGL((label_attn_bwd_bf16_kernel<false>), grid, block, T::total, st, a);
    GZSL_CHECK_LAUNCH();
    if (defer_dq_reduce) return 0;
    return gzsl_label_attn_bwd_bf16x6_dq_reduce(work, dQ, B, Tp, E, F, L, stream);
}
