// EXPERIMENTAL, opt-in (GZSL_ATTN_BF16X6=1 on the host side; never the default): the forward label attention of attn.hip
// with every fp32 product formed on the bf16 matrix pipe.
//
// An fp32 value is exactly the sum of three bf16 pieces, a = h + m + l (8 significant bits each).  A product a.b is
// taken as the six cross terms of weight >= 2^-16 (hh hm mh mm hl lh; the dropped ml lm ll are <= 2^-23 relative),
// each one v_mfma_f32_16x16x32_bf16 with fp32 accumulation: 6 MFMAs of 16 cycles per 32 k against 8 MFMAs of 32 cycles
// with v_mfma_f32_16x16x4_f32.  tools/bf16_split_gemm.hip measures the error of this form against fp64 (it is the
// accumulation's, not the split's: 4.8e-7 rms relative against 5.7e-7 for an fp32 FMA chain).
//
// Same contract as gzsl_label_attn_fwd (icd_modules.py:137-142; same dropout hash, same online softmax).  Differences in
// structure: keys and values are first repacked (attn_pack_bf16_kernel) into per-32-token tiles that already ARE the
// LDS images -- three bf16 planes of K [32 tokens][EK] and of V^T [FP][32 tokens], 16-byte chunks XOR-swizzled so that
// every ds_read_b128 below is conflict-free -- so staging is a flat 16-byte copy; a workgroup is 8 waves = 128 labels
// (half the tile traffic per label of the fp32 kernel's 64); P is split into its three planes in registers.
#include "common.h"
#include "bf16x3.h"
#include "rng.h"
#include "../../include/gzsl_hip.h"

#define BF_BT 32        // tokens per tile = k of one MFMA
#define BF_WAVES 8      // 128 labels per workgroup
#define BF_RESCALE 12.0f

// Tile image (bytes): K planes [3][32 tokens][EK bf16], chunk c (8 e) of token row r at chunk c ^ (r & 15);
//                     V^T planes [3][FP f][32 tokens bf16], chunk q of row f at chunk q ^ ((f >> 1) & 3), where chunk q
//                     holds the k slots (q, j) <-> tokens 16 (j >> 2) + 4 q + (j & 3): the order in which a lane of the
//                     S^T accumulator holds its eight tokens of the tile.
template <int EKS, int NFT>
struct BfTile {
    static constexpr int EK = 32 * EKS, FP = 16 * NFT;
    static constexpr int PLK = BF_BT * EK * 2, PLV = FP * BF_BT * 2;
    static constexpr int TK = 3 * PLK, TV = 3 * PLV, bytes = TK + TV, u4 = bytes / 16;
};

template <int EKS, int NFT>
__global__ __launch_bounds__(256) void attn_pack_bf16_kernel(const float* __restrict__ PX, const float* __restrict__ X,
                                                             unsigned char* __restrict__ work, int Tp, int E, int F) {
    using T = BfTile<EKS, NFT>;
    const int tile = blockIdx.x, b = blockIdx.y, nT = gridDim.x;
    unsigned char* out = work + ((long long)b * nT + tile) * T::bytes;
    const float* PXb = PX + (long long)b * Tp * E;
    const float* Xb = X + (long long)b * Tp * F;
    const int t0 = tile * BF_BT;
    // keys: item = (token row r, chunk c); consecutive threads take consecutive chunks of a row
    constexpr int KCH = T::EK / 8;
    for (int it = threadIdx.x; it < BF_BT * KCH; it += blockDim.x) {
        const int r = it / KCH, c = it % KCH;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int e = 8 * c + j, t = t0 + r;
            v[j] = (t < Tp && e < E) ? PXb[(long long)t * E + e] : 0.f;
        }
        bf16x8 pl[3];
        split3_x8(v, pl);
#pragma unroll
        for (int p = 0; p < 3; ++p)
            *reinterpret_cast<bf16x8*>(out + p * T::PLK + r * (T::EK * 2) + ((c ^ (r & 15)) * 16)) = pl[p];
    }
    // values, transposed: item = (chunk q, row f); consecutive threads take consecutive f (coalesced reads of a token row)
    for (int it = threadIdx.x; it < 4 * T::FP; it += blockDim.x) {
        const int q = it / T::FP, f = it % T::FP;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int t = t0 + 16 * (j >> 2) + 4 * q + (j & 3);
            v[j] = (t < Tp && f < F) ? Xb[(long long)t * F + f] : 0.f;
        }
        bf16x8 pl[3];
        split3_x8(v, pl);
#pragma unroll
        for (int p = 0; p < 3; ++p)
            *reinterpret_cast<bf16x8*>(out + T::TK + p * T::PLV + f * 64 + ((q ^ ((f >> 1) & 3)) * 16)) = pl[p];
    }
}

struct AttnFwdBfArgs {
    const unsigned char* work;  // packed tiles [B][nT][tile bytes]
    const float* Q;
    float* O;
    float* LSE;
    int B, Tp, E, F, L;
    int l_offset, L_total;
    uint32_t seed, thresh;
    float scale;
};

template <int EKS, int NFT, bool DROP>
__global__ __launch_bounds__(BF_WAVES * 64) void label_attn_fwd_bf16_kernel(AttnFwdBfArgs a) {
    using T = BfTile<EKS, NFT>;
    constexpr int NT = BF_WAVES * 64;
    constexpr int NS = (T::u4 + NT - 1) / NT;  // 16-byte staging slots per thread
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_bf[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 15, lq = lane >> 4;
    const int nLT = (a.L + 16 * BF_WAVES - 1) / (16 * BF_WAVES);
    const int nT = (a.Tp + BF_BT - 1) / BF_BT;
    int b, lt;
    {
        const int id = blockIdx.x;
        if ((a.B & 7) == 0) {  // same-XCD workgroups share a note (its tiles stay in that L2)
            const int xcd = id & 7, slot = id >> 3;
            b = (slot / nLT) * 8 + xcd;
            lt = slot % nLT;
        } else {
            b = id / nLT;
            lt = id % nLT;
        }
    }
    const int lbl = lt * 16 * BF_WAVES + wave * 16 + lr;
    const bool lbl_ok = lbl < a.L;

    // Q planes, B operand of S^T = K.Q^T: slot j of k step ks is e = 32 ks + 8 lq + j
    bf16x8 qf[EKS][3];
    {
        const float* qrow = a.Q + (long long)min(lbl, a.L - 1) * a.E;
#pragma unroll
        for (int ks = 0; ks < EKS; ++ks) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int e = 32 * ks + 8 * lq + j;
                v[j] = qrow[min(e, a.E - 1)] * ((lbl_ok && e < a.E) ? 1.f : 0.f);
            }
            split3_x8(v, qf[ks]);
        }
    }
    const uint32_t rowkey = gz_row_key(a.seed, GZSL_STREAM_ATTN, (uint32_t)b * (uint32_t)a.L_total + (uint32_t)(a.l_offset + lbl));
    uint32_t colmul[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) colmul[r] = (uint32_t)(4 * lq + r) * 0x85EBCA6Bu;
    f32x4 acc[NFT];
#pragma unroll
    for (int j = 0; j < NFT; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m = -INFINITY, lsum = 0.f;

    const u32x4* gt = reinterpret_cast<const u32x4*>(a.work + (long long)b * nT * T::bytes);
    u32x4 sreg[NS];
    auto issue = [&](int tile) {
        const u32x4* g = gt + (long long)tile * T::u4;
#pragma unroll
        for (int i = 0; i < NS; ++i) sreg[i] = g[min(tid + NT * i, T::u4 - 1)];
    };
    // Where two tile images fit LDS (E <= 128; one workgroup per CU either way) the tiles alternate between two
    // buffers with ONE barrier per tile: iteration t stores tile t+1 (in registers since iteration t-1) into the other
    // buffer, issues the loads of tile t+2, computes tile t.  Otherwise one buffer, stored between two barriers.
    constexpr int NBUF = (2 * T::bytes <= 160 * 1024) ? 2 : 1;
    auto commit = [&](int buf) {
        unsigned char* dst = smem_bf + buf * T::bytes;
#pragma unroll
        for (int i = 0; i < NS; ++i)
            if (tid + NT * i < T::u4) reinterpret_cast<u32x4*>(dst)[tid + NT * i] = sreg[i];
    };
    const int vsw = (lq ^ ((lr >> 1) & 3)) * 16;  // swizzled chunk of row f = 16 j + lr ((f >> 1) & 3 = (lr >> 1) & 3)

    issue(0);
    commit(0);
    if (nT > 1) issue(1);
    __syncthreads();
    for (int tile = 0; tile < nT; ++tile) {
        const int t0 = tile * BF_BT;
        const unsigned char* Ks = smem_bf + (NBUF == 2 ? (tile & 1) : 0) * T::bytes;
        const unsigned char* Vs = Ks + T::TK;
        if (NBUF == 2) {
            if (tile + 1 < nT) commit((tile + 1) & 1);
            if (tile + 2 < nT) issue(tile + 2);
        }
        // S^T sub-tiles: rows = tokens t0 + 16u + 4lq + r, cols = labels.  Key fragments are read one (u, ks) item ahead.
        float sv[8];
        {
            bf16x8 ka[2][3];
            auto ldk = [&](int item, int slot) {
                const int u = item / EKS, ks = item % EKS;
                const unsigned char* krow = Ks + (16 * u + lr) * (T::EK * 2) + (((4 * ks + lq) ^ lr) * 16);
#pragma unroll
                for (int p = 0; p < 3; ++p) ka[slot][p] = *reinterpret_cast<const bf16x8*>(krow + p * T::PLK);
            };
            constexpr bool AHEAD = EKS <= 4;  // E > 128 has no registers left for the second fragment set
            if (AHEAD) ldk(0, 0);
            f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int item = 0; item < 2 * EKS; ++item) {
                if (!AHEAD) ldk(item, 0);
                else if (item + 1 < 2 * EKS) ldk(item + 1, (item + 1) & 1);
                if (AHEAD) __builtin_amdgcn_sched_barrier(0);
                s = mfma_x6(ka[AHEAD ? (item & 1) : 0], qf[item % EKS], s);
                if (AHEAD) __builtin_amdgcn_sched_barrier(0);
                if (item % EKS == EKS - 1) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) sv[4 * (item / EKS) + r] = s[r];
                    s = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
        }
        if (t0 + BF_BT > a.Tp) {  // last tile: rows past the end of the note
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (t0 + 16 * (i >> 2) + 4 * lq + (i & 3) >= a.Tp) sv[i] = -INFINITY;
        }
        float mt = -INFINITY;
#pragma unroll
        for (int i = 0; i < 8; ++i) mt = fmaxf(mt, sv[i]);
        if (__any(mt > m + BF_RESCALE)) {  // lazy online-softmax rescale (as attn.hip)
            mt = fmaxf(mt, __shfl_xor(mt, 16));
            mt = fmaxf(mt, __shfl_xor(mt, 32));
            const float mn = fmaxf(m, mt);
            const float alpha = __expf(m - mn);
#pragma unroll
            for (int j = 0; j < NFT; ++j) acc[j] *= alpha;
            lsum *= alpha;
            m = mn;
        }
        // P, dropout, and its three planes: k slot j = 4u + r <-> token 16u + 4lq + r (the order of the packed V^T chunks)
        bf16x8 pb[3];
        {
            float pv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float p = __expf(sv[i] - m);
                lsum += p;
                if (DROP) {
                    const uint32_t uu = gz_mix32(rowkey ^ (colmul[i & 3] + (uint32_t)(t0 + 16 * (i >> 2)) * 0x85EBCA6Bu));
                    p = (uu >= a.thresh) ? p * a.scale : 0.f;
                }
                pv[i] = p;
            }
            split3_x8(pv, pb);
        }
        // O^T[f, label] += V^T[f, tokens] . P^T[tokens, label]; value fragments read one tile of f ahead
        {
            bf16x8 va[2][3];
            auto ldv = [&](int j, int slot) {
                const unsigned char* vrow = Vs + (16 * j + lr) * 64 + vsw;
#pragma unroll
                for (int p = 0; p < 3; ++p) va[slot][p] = *reinterpret_cast<const bf16x8*>(vrow + p * T::PLV);
            };
            constexpr bool AHEAD = EKS <= 4;
            if (AHEAD) ldv(0, 0);
#pragma unroll
            for (int j = 0; j < NFT; ++j) {
                if (!AHEAD) ldv(j, 0);
                else if (j + 1 < NFT) ldv(j + 1, (j + 1) & 1);
                if (AHEAD) __builtin_amdgcn_sched_barrier(0);
                acc[j] = mfma_x6(va[AHEAD ? (j & 1) : 0], pb, acc[j]);
                if (AHEAD) __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();  // tile t consumed by every wave (and, with two buffers, tile t+1 stored by every wave)
        if (NBUF == 1) {
            if (tile + 1 < nT) commit(0);
            if (tile + 2 < nT) issue(tile + 2);
            __syncthreads();
        }
    }
    lsum += __shfl_xor(lsum, 16);
    lsum += __shfl_xor(lsum, 32);
    if (lbl_ok) {
        const float inv = 1.0f / lsum;
        if (lq == 0) a.LSE[(long long)b * a.L + lbl] = m + __logf(lsum);
        float* Op = a.O + ((long long)b * a.L + lbl) * a.F;
        const bool vec = (a.F & 3) == 0 && (((uintptr_t)a.O) & 15) == 0;
        // acc[j][r] = O[lbl][16j + 4lq + r]
#pragma unroll
        for (int j = 0; j < NFT; ++j) {
            const int f = 16 * j + 4 * lq;
            if (vec && f + 3 < a.F) {
                *reinterpret_cast<float4*>(Op + f) = make_float4(acc[j][0] * inv, acc[j][1] * inv, acc[j][2] * inv, acc[j][3] * inv);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (f + r < a.F) Op[f + r] = acc[j][r] * inv;
            }
        }
    }
}

template <int EKS, int NFT>
static int launch_bf(const float* PX, const float* X, const AttnFwdBfArgs& a, unsigned char* work, hipStream_t st) {
    using T = BfTile<EKS, NFT>;
    static_assert(T::bytes % 16 == 0 && T::bytes <= 160 * 1024, "tile image");
    const int nT = cdiv(a.Tp, BF_BT), nLT = cdiv(a.L, 16 * BF_WAVES);
    hipLaunchKernelGGL((attn_pack_bf16_kernel<EKS, NFT>), dim3(nT, a.B), dim3(256), 0, st, PX, X, work, a.Tp, a.E, a.F);
    GZSL_CHECK_LAUNCH();
    if (a.thresh)
        hipLaunchKernelGGL((label_attn_fwd_bf16_kernel<EKS, NFT, true>), dim3(a.B * nLT), dim3(BF_WAVES * 64), (2 * T::bytes <= 160 * 1024 ? 2 : 1) * T::bytes, st, a);
    else
        hipLaunchKernelGGL((label_attn_fwd_bf16_kernel<EKS, NFT, false>), dim3(a.B * nLT), dim3(BF_WAVES * 64), (2 * T::bytes <= 160 * 1024 ? 2 : 1) * T::bytes, st, a);
    GZSL_CHECK_LAUNCH();
    return 0;
}

static long long bf_tile_bytes(int E, int F) {
    if (F > 208 || E > 256 || E <= 0 || F <= 0) return 0;
    return E <= 128 ? BfTile<4, 13>::bytes : BfTile<8, 13>::bytes;
}

extern "C" long long gzsl_label_attn_fwd_bf16x6_work_bytes(int B, int Tp, int E, int F) {
    if (B <= 0 || Tp <= 0) return 0;
    return (long long)B * cdiv(Tp, BF_BT) * bf_tile_bytes(E, F);
}

extern "C" int gzsl_label_attn_fwd_bf16x6(const float* PX, const float* X, const float* Q, float* O, float* LSE, int B,
                                          int Tp, int E, int F, int L, int l_offset, int L_total, uint32_t seed,
                                          uint32_t drop_thresh, float drop_scale, void* work, long long work_bytes,
                                          void* stream) {
    if (B <= 0 || L <= 0) return 0;
    if (Tp <= 0 || E <= 0 || F <= 0 || !work || (((uintptr_t)work) & 15) != 0) return GZSL_EINVAL;
    const long long need = gzsl_label_attn_fwd_bf16x6_work_bytes(B, Tp, E, F);
    if (need == 0 || work_bytes < need) return GZSL_EINVAL;
    AttnFwdBfArgs a{(const unsigned char*)work, Q, O, LSE, B, Tp, E, F, L, l_offset, L_total, seed, drop_thresh, drop_scale};
    hipStream_t st = (hipStream_t)stream;
    if (E <= 128) return launch_bf<4, 13>(PX, X, a, (unsigned char*)work, st);
    return launch_bf<8, 13>(PX, X, a, (unsigned char*)work, st);
}
