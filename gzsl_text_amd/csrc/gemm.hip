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
// Design (round-1 v3):
//   * 256 threads = 4 waves stacked along M, each wave owns WTM x WTN tiles of v_mfma_f32_16x16x4_f32 (exact fp32,
//     64 FLOP/clk/SIMD: the only fp32 matrix rate gfx950 has; tools/mfma_peak.hip measures 156.3 TFLOP/s sustained).
//     Tiles are small (64 or 128 rows x 112/128/208 columns) and register use is bounded so that 2-3 workgroups
//     share a CU: their barriers and global-load waits overlap each other's MFMA phases, and the many small tiles
//     keep the last round of the grid short.
//   * BK = 16, two LDS buffers, ONE barrier per k-tile.  Tile t+1 travels
//     global -> registers while tile t's MFMAs issue, then registers -> the other LDS buffer.
//   * Every operand moves as 16-byte vectors end to end.  An operand that is contiguous along k ("KC") is staged
//     as [row][k] and a lane's ds_read_b128 holds its k values for FOUR consecutive MFMA steps (k = 16*group +
//     4*quarter + step: the order of the k sum is free as long as A and B agree).  An operand that is contiguous
//     along its rows ("RC": weight-gradient products, W in data-gradient products) is staged as [k][row] and tile i
//     of a lane covers row 64*(i/4) + 4*lane + i%4, so that one ds_read_b128 holds one k step of FOUR tiles.  No
//     transposes through LDS, no scalar LDS traffic, no bank conflicts (see kc_off).
//   * Bounds / window / k-range predicates never branch around a load: the address is clamped to the operand base
//     and the loaded bits are ANDed with a 0/~0 mask, so all loads of a tile stay in flight together.
//   * Operands whose strides or extents are not multiples of 4 floats use the same layouts with scalar loads
//     (VEC = false instantiations).
#include "common.h"
#include "../../include/gzsl_hip.h"

#ifndef BK
#define BK 16  // 16: 64-byte rows, 3 workgroups per CU; 32: full 128-byte lines, 2 per CU (measured equal within 4 %)
#endif
#define LDK BK  // [row][k] images are unpadded; see kc_off
#define NTHREADS 256
#ifndef GZ_GEMM_ABL
#define GZ_GEMM_ABL 0  // timing experiments only (tools/ablate_gemm.sh): 1 no loads, 2 no LDS stores, 4 no barrier, 8 no MFMA
#endif
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct Operand {
    const float* p;
    long long mbs, kbs;  // strides between row batches / k batches
    int rows, mb, sm, sk, shift;
    unsigned span;
};

// [row][k] image: the 16-byte chunk c (k = 4c..4c+3) of row r lives at chunk c ^ swz(r).  ds_read_b128 serves the lane
// groups {0-3,12-15,20-27}, {4-11,16-19,28-31} (+32) with banks (a/4) mod 64 (MI355X_MICROARCH.md, LDS): a lane of
// quarter q reads chunk 4*kgroup + q of row 16*tile + (lane & 15), and with these XORs each group's 16 lanes cover
// all 64 banks exactly once (4 LDS cycles, the minimum); the 8-lane groups of ds_write_b128 are conflict-free too.
__device__ __forceinline__ int kc_off(int r, int c) {
#if BK == 32
    return r * LDK + 4 * (c ^ ((r >> 1) & 7));  // 128-byte rows: two rows per bank row
#else
    return r * LDK + 4 * (c ^ ((r >> 2) & 2));  // 64-byte rows: four rows per bank row
#endif
}

__device__ __forceinline__ float mask_bits(float v, unsigned m) { return __uint_as_float(__float_as_uint(v) & m); }

// Global -> register -> LDS mover for one operand tile of ROWS rows x BK k.  RPAD = row count of the [k][row] image.
template <int ROWS, int RPAD, bool RC, bool VEC>
struct Mover {
    static constexpr int TOTAL = ROWS * BK / 4;  // 16-byte slots in the tile
    static constexpr int NS = (TOTAL + NTHREADS - 1) / NTHREADS;
    static constexpr int NR = (RC && !VEC) ? 4 : 1;  // distinct rows per slot that need their own state
    long long rb[NS][NR];
    int orow[NS][NR];
    int kc[NS], lds_off[NS];
    unsigned coff[NS];  // fast path: float offset of the slot from (operand base + k0 * sk); rows past the end -> row 0
    struct Tile {  // one tile's worth of this thread's slots, in flight or landed
        unsigned msk[NS][VEC ? 1 : 4];
        f32x4 v[NS];
    };

    // dump: float offset (from the operand's image in an LDS buffer) of this thread's 16-byte parking place: a thread
    // without a slot stores there, so that the stores need no branch
    __device__ __forceinline__ void init(const Operand& op, int tid, int row0, int dump) {
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            int s = tid + NTHREADS * i;
            bool st = s < TOTAL;
            int sc = st ? s : TOTAL - 1;
            int r;
            if (!RC) {
                r = sc / (BK / 4);
                kc[i] = 4 * (sc % (BK / 4));
                lds_off[i] = st ? kc_off(r, kc[i] >> 2) : dump;
            } else {
                kc[i] = sc / (ROWS / 4);
                r = 4 * (sc % (ROWS / 4));
                lds_off[i] = st ? kc[i] * RPAD + r : dump;
            }
#pragma unroll
            for (int e = 0; e < NR; ++e) {
                int row = row0 + r + e;
                bool valid = row < op.rows;
                int rowc = valid ? row : 0;
                int mbi = rowc / op.mb;
                int mi = rowc - mbi * op.mb;
                rb[i][e] = (long long)mbi * op.mbs;
                // a row outside the operand gets an offset that no k can bring back into [0, span)
                orow[i][e] = valid ? mi * op.sm + op.shift : (int)0x80000000;
                if (e == 0) coff[i] = (unsigned)(rb[i][0] + mi * op.sm + op.shift + kc[i] * op.sk);
            }
        }
    }

    // k0: first k of the tile; (kb0, ki0): its batch decomposition
    __device__ __forceinline__ void load(const Operand& op, int k0, int kb0, int ki0, int kend, int k_inner,
                                         Tile& d) const {
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            if (VEC) {
                int ki = ki0 + kc[i], kb = kb0;
                if (ki >= k_inner) {
                    ki -= k_inner;
                    kb += 1;
                }
                int o = orow[i][0] + ki * op.sk;
                bool ok = (k0 + kc[i] < kend) && ((unsigned)o < op.span);
                unsigned m = ok ? 0xFFFFFFFFu : 0u;
                long long off = (rb[i][0] + (long long)kb * op.kbs + o) & (long long)(int)m;  // no branch
                d.v[i] = *reinterpret_cast<const f32x4*>(op.p + off);
                d.msk[i][0] = m;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int er = RC ? e : 0;
                    int kk = kc[i] + (RC ? 0 : e);
                    int ki = ki0 + kk, kb = kb0;
                    if (ki >= k_inner) {
                        ki -= k_inner;
                        kb += 1;
                    }
                    int o = orow[i][er] + ki * op.sk;
                    bool ok = (k0 + kk < kend) && ((unsigned)o < op.span);
                    unsigned m = ok ? 0xFFFFFFFFu : 0u;
                    long long off = (rb[i][er] + (long long)kb * op.kbs + o) & (long long)(int)m;
                    d.v[i][e] = op.p[off];
                    d.msk[i][e] = m;
                }
            }
        }
    }

    // Fast path (whole k-tile inside K, no window, no k batches): tb = operand base + k0 * sk is wave-uniform, the
    // per-slot offset is a constant, nothing is masked.  Rows past the operand's end read row 0 instead: they only
    // feed output rows / columns that are never stored.  (fp32 MFMA and VALU do not overlap on a SIMD, so every
    // vector instruction of the address generation is paid in matrix time: this path has one add per slot.)
    __device__ __forceinline__ void load_fast(const float* tb, Tile& d) const {
#pragma unroll
        for (int i = 0; i < NS; ++i) d.v[i] = *reinterpret_cast<const f32x4*>(tb + coff[i]);
    }
    __device__ __forceinline__ void store_fast(float* buf, const Tile& d) const {
#pragma unroll
        for (int i = 0; i < NS; ++i) *reinterpret_cast<f32x4*>(buf + lds_off[i]) = d.v[i];
    }

    // the mask is applied here, after the tile's MFMAs, so that nothing waits on a load before it has to
    __device__ __forceinline__ void store(float* buf, const Tile& d) const {
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            f32x4 t = d.v[i];
#pragma unroll
            for (int e = 0; e < 4; ++e) t[e] = mask_bits(t[e], d.msk[i][VEC ? 0 : e]);
            *reinterpret_cast<f32x4*>(buf + lds_off[i]) = t;
        }
    }
};

// tile / lane -> row of an operand block of NTOT tiles.  [row][k] images use the natural order.  [k][row] images
// are read as vectors along the rows, so tiles are grouped: a group of 4 tiles covers 64 rows with tile e of lane l
// on row 4*l + e; what is left of NTOT forms a group of 2 (row 2*l + e) and / or a single natural tile.
template <bool RC, int NTOT>
__device__ __forceinline__ int tile_row(int tile, int l) {
    if (!RC) return 16 * tile + l;
    constexpr int G4 = NTOT / 4 * 4;
    constexpr bool HAS2 = (NTOT - G4) >= 2;
    if (tile < G4) return 64 * (tile >> 2) + 4 * l + (tile & 3);
    if (HAS2 && tile < G4 + 2) return 16 * G4 + 2 * l + (tile - G4);
    return 16 * (G4 + (HAS2 ? 2 : 0)) + l;
}

// Fragments f[tile][step] of NT tiles for the four k steps of the lane's quarter in k group g (k = 16g + 4*lq + step).
// Either NT == NTOT (the whole block, t0 == 0) or NT in {1, 2} and NTOT is a multiple of 4 (a wave's share of A).
template <int NT, int NTOT, int RPAD, bool RC>
__device__ __forceinline__ void read_frags(const float* buf, int g, int t0, int lr, int lq, float (&f)[NT][4]) {
    if (!RC) {
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            f32x4 t = *reinterpret_cast<const f32x4*>(buf + kc_off(16 * (t0 + i) + lr, 4 * g + lq));
#pragma unroll
            for (int j = 0; j < 4; ++j) f[i][j] = t[j];
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float* row = buf + (16 * g + 4 * lq + j) * RPAD;
            if (NT != NTOT && NT == 1) {
                f[0][j] = row[64 * (t0 >> 2) + 4 * lr + (t0 & 3)];
            } else if (NT != NTOT && NT == 2) {
                f32x2 t = *reinterpret_cast<const f32x2*>(row + 64 * (t0 >> 2) + 4 * lr + (t0 & 3));
                f[0][j] = t[0];
                f[1][j] = t[1];
            } else {
                constexpr int G4 = NT / 4 * 4;
                constexpr bool HAS2 = (NT - G4) >= 2;
                constexpr bool HAS1 = ((NT - G4) & 1) != 0;
#pragma unroll
                for (int gi = 0; gi < NT / 4; ++gi) {
                    f32x4 t = *reinterpret_cast<const f32x4*>(row + 64 * gi + 4 * lr);
#pragma unroll
                    for (int e = 0; e < 4; ++e) f[4 * gi + e][j] = t[e];
                }
                if (HAS2) {
                    f32x2 t = *reinterpret_cast<const f32x2*>(row + 16 * G4 + 2 * lr);
                    f[G4][j] = t[0];
                    f[G4 + 1][j] = t[1];
                }
                if (HAS1) f[NT - 1][j] = row[16 * (G4 + (HAS2 ? 2 : 0)) + lr];
            }
        }
    }
}

template <int WTM, int WTN, bool A_RC, bool B_RC, bool VEC>
__global__ __launch_bounds__(NTHREADS, !VEC ? 1 : (WTM == 1 ? (BK == 16 ? 3 : 2) : (BK == 16 ? 2 : 1))) void gemm_f32_kernel(GzslGemm g) {
    constexpr int BM = 64 * WTM;
    constexpr int BN = 16 * WTN;
    constexpr int BNP = BN;  // [k][row] image of B: rows in natural order, BN is a multiple of 16 (bank rule)
    constexpr int ASZ = A_RC ? BK * BM : BM * LDK;
    constexpr int BSZ = B_RC ? BK * BNP : BN * LDK;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* lds = reinterpret_cast<float*>(smem_raw);  // [2][ASZ + BSZ + 4 * NTHREADS (parking area)]
    constexpr int BUF = ASZ + BSZ + 4 * NTHREADS;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int lr = lane & 15;
    const int lq = lane >> 4;
    const int tiles_n = (g.N + BN - 1) / BN;
    const int m0 = (blockIdx.x / tiles_n) * BM;
    const int n0 = (blockIdx.x % tiles_n) * BN;
    const int z = blockIdx.z;
    int kbeg = 0, kend = g.K;
    if (g.split_k > 1) {
        kbeg = z * g.k_per_split;
        kend = min(g.K, kbeg + g.k_per_split);
    }

    Operand oa{g.A, g.a_mbs, g.a_kbs, g.M, g.a_mb, g.a_sm, g.a_sk, g.a_shift, g.a_span};
    Operand ob{g.B, 0ll, g.b_kbs, g.N, 0x7FFFFFFF, g.b_sn, g.b_sk, 0, 0x7FFFFFFFu};
    Mover<BM, BM, A_RC, VEC> ma;
    Mover<BN, BNP, B_RC, VEC> mb;
    ma.init(oa, tid, m0, ASZ + BSZ + 4 * tid);
    mb.init(ob, tid, n0, BSZ + 4 * tid);

    f32x4 acc[WTM][WTN];
#pragma unroll
    for (int i = 0; i < WTM; ++i)
#pragma unroll
        for (int j = 0; j < WTN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int k_inner = g.k_inner;
    const int ntile = (kend - kbeg + BK - 1) / BK;
    // (kb, ki): batch decomposition of the first k of the next tile to fetch; k_next: that k
    int k_next = kbeg;
    int kb0 = kbeg / k_inner;
    int ki0 = kbeg - kb0 * k_inner;
    typename Mover<BM, BM, A_RC, VEC>::Tile ga;
    typename Mover<BN, BNP, B_RC, VEC>::Tile gb;
    // A k-tile takes the fast path (see Mover::load_fast) when it lies wholly inside K and inside ONE k batch, and the
    // workgroup's rows need no window test: either the operand has no window, or (conv data gradient) all BM rows are
    // real rows of one note whose windows lie inside the note for every k -- 36 of a note's 39 row tiles; the tiles at the
    // note boundaries, the K tail and k-tiles that straddle two k batches (conv weight gradient: 1 in 155) are masked.
    bool rows_fast = VEC && g.a_span == 0x7FFFFFFFu && g.a_shift == 0;
    if (VEC && !rows_fast && g.a_sm >= 0 && g.a_sk >= 0) {
        const int mlast = m0 + BM - 1;
        if (mlast < g.M && m0 / g.a_mb == mlast / g.a_mb) {
            const long long lo = (long long)(m0 % g.a_mb) * g.a_sm + g.a_shift;
            const long long hi = (long long)(mlast % g.a_mb) * g.a_sm + g.a_shift +
                                 (long long)(min(g.K, k_inner) - 1) * g.a_sk + (A_RC ? 0 : 3);
            rows_fast = lo >= 0 && hi < (long long)g.a_span;
        }
    }
    bool in_regs_fast = false;  // which path fetched the tile now in registers
    auto fetch = [&](int t) {   // tile t -> registers (t >= ntile: masked to zeros, addresses clamped: harmless)
        const bool tf = rows_fast && k_next + BK <= kend && ki0 + BK <= k_inner;
        if (!(GZ_GEMM_ABL & 1)) {
            if (tf) {
                ma.load_fast(oa.p + (long long)kb0 * oa.kbs + (long long)ki0 * oa.sk, ga);
                mb.load_fast(ob.p + (long long)kb0 * ob.kbs + (long long)ki0 * ob.sk, gb);
            } else {
                ma.load(oa, k_next, kb0, ki0, kend, k_inner, ga);
                mb.load(ob, k_next, kb0, ki0, kend, k_inner, gb);
            }
        }
        in_regs_fast = tf;
        k_next += BK;
        ki0 += BK;
        if (ki0 >= k_inner) {
            ki0 -= k_inner;
            kb0 += 1;
        }
    };
    auto stage = [&](float* buf) {  // registers -> LDS buffer
        if (GZ_GEMM_ABL & 2) return;
        if (in_regs_fast) {
            ma.store_fast(buf, ga);
            mb.store_fast(buf + ASZ, gb);
        } else {
            ma.store(buf, ga);
            mb.store(buf + ASZ, gb);
        }
    };
    fetch(0);
    stage(lds);
    __syncthreads();
    int cur = 0;
    for (int t = 0; t < ntile; ++t) {
        fetch(t + 1);                       // tile t+1 -> registers
        __builtin_amdgcn_sched_barrier(0);  // every load is in flight before tile t's MFMAs start
        const float* abuf = lds + cur * BUF;
#pragma unroll
        for (int kg = 0; kg < BK / 16; ++kg) {
            if (kg > 0 && kbeg + t * BK + 16 * kg >= kend) break;  // K tail: nothing but zeros in this k group
            float fa[WTM][4];
            read_frags<WTM, 4 * WTM, BM, A_RC>(abuf, kg, wave * WTM, lr, lq, fa);
            if constexpr (!B_RC && WTN > 8 && !(GZ_GEMM_ABL & 8)) {
                // [row][k] image of a wide B tile: its fragments are read and consumed in two halves.  All 13 x 4 at once
                // put the kernel over the 168 registers three workgroups per CU allow, and the spill reloads sat inside
                // this loop -- behind a vmcnt wait that also waited for the tile loads issued at the top of the iteration
                // (the whole global-load latency exposed, every k-tile).
                constexpr int H0 = (WTN + 1) / 2, H1 = WTN - H0;
                const float* bb = abuf + ASZ;
                float fb0[H0][4], fb1[H1][4];
#pragma unroll
                for (int i = 0; i < H0; ++i) {
                    const f32x4 t = *reinterpret_cast<const f32x4*>(bb + kc_off(16 * i + lr, 4 * kg + lq));
#pragma unroll
                    for (int j = 0; j < 4; ++j) fb0[i][j] = t[j];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int i = 0; i < WTM; ++i)
#pragma unroll
                        for (int n = 0; n < H0; ++n) acc[i][n] = mfma16(fa[i][j], fb0[n][j], acc[i][n]);
#pragma unroll
                for (int i = 0; i < H1; ++i) {
                    const f32x4 t = *reinterpret_cast<const f32x4*>(bb + kc_off(16 * (H0 + i) + lr, 4 * kg + lq));
#pragma unroll
                    for (int j = 0; j < 4; ++j) fb1[i][j] = t[j];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int i = 0; i < WTM; ++i)
#pragma unroll
                        for (int n = 0; n < H1; ++n) acc[i][H0 + n] = mfma16(fa[i][j], fb1[n][j], acc[i][H0 + n]);
            } else {
                float fb[WTN][4];
                read_frags<WTN, WTN, BNP, B_RC>(abuf + ASZ, kg, 0, lr, lq, fb);
                if (!(GZ_GEMM_ABL & 8)) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int i = 0; i < WTM; ++i)
#pragma unroll
                            for (int n = 0; n < WTN; ++n) acc[i][n] = mfma16(fa[i][j], fb[n][j], acc[i][n]);
                } else {
                    acc[0][0][0] += fa[0][0] + fb[0][0];
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);  // ... and nothing waits on them until the MFMAs have issued
        cur ^= 1;
        stage(lds + cur * BUF);
        if (!(GZ_GEMM_ABL & 4)) __syncthreads();
    }

    float* C = g.C + (g.split_k > 1 ? (long long)z * g.slab_stride : 0);
    const bool plain = g.split_k > 1;
    // Row-contiguous epilogue: the wave's 16 x (16*WTN) block goes through its private LDS rows (the k loop's last
    // barrier has retired every read of the tile buffers) and leaves as 16-byte stores, 128 columns per pass, instead
    // of one 4-byte store per accumulator register (4 rows x 64 B per instruction).
    if (WTM == 1 && ((g.N | g.ldc) & 3) == 0 && (((uintptr_t)C | (uintptr_t)g.bias) & 15) == 0) {
        constexpr int STW = 132;
        float* st = lds + wave * (16 * STW);
        const int rr0 = lane >> 5, c4 = (lane & 31) * 4;
        float dotp[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // fused head: partial dot products of rows 2i + rr0
#pragma unroll
        for (int ps = 0; ps < (16 * WTN + 127) / 128; ++ps) {
#pragma unroll
            for (int j = 0; j < WTN; ++j) {
                if (tile_row<B_RC, WTN>(j, 0) / 128 == ps) {
                    const int c = tile_row<B_RC, WTN>(j, lr) - 128 * ps;
#pragma unroll
                    for (int r = 0; r < 4; ++r) st[(4 * lq + r) * STW + c] = acc[0][j][r];
                }
            }
            wave_lds_fence();  // the rows read below were staged by other lanes of this wave
            const int col = n0 + 128 * ps + c4;
            const bool col_ok = (128 * ps + c4 < 16 * WTN) && col < g.N;
            f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f};
            if (!plain && g.bias && col_ok) bv = *reinterpret_cast<const f32x4*>(g.bias + col);
#pragma unroll
            for (int r2 = 0; r2 < 16; r2 += 2) {
                const int rr = r2 + rr0;
                const int row = m0 + tile_row<A_RC, 4>(wave, rr);
                if (!col_ok || row >= g.M) continue;
                f32x4 v = *reinterpret_cast<const f32x4*>(st + rr * STW + c4);
                float* cp = C + (long long)row * g.ldc + col;
                v *= g.alpha;
                if (!plain) {
                    v += bv;
                    if (g.accumulate) v += *reinterpret_cast<const f32x4*>(cp);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (g.act == GZSL_ACT_RELU) v[e] = fmaxf(v[e], 0.f);
                        else if (g.act == GZSL_ACT_TANH) v[e] = tanhf(v[e]);
                        else if (g.act == GZSL_ACT_LRELU) v[e] = v[e] > 0.f ? v[e] : 0.2f * v[e];
                    }
                }
                if (g.dot_w) {  // fused head: this lane's four columns of <relu(row), dot_w[row % period]>
                    const f32x4 w = *reinterpret_cast<const f32x4*>(g.dot_w + (long long)(row % g.dot_period) * g.N + col);
#pragma unroll
                    for (int e = 0; e < 4; ++e) dotp[r2 >> 1] += fmaxf(v[e], 0.f) * w[e];
                    if (g.dot_no_store) continue;
                }
                *reinterpret_cast<f32x4*>(cp) = v;
            }
            wave_lds_fence();  // ... and the next pass overwrites them
        }
        if (g.dot_w) {  // a row's columns sit in the 32 lanes of one half-wave (rr0 = lane >> 5): butterfly inside it
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float s = dotp[i];
#pragma unroll
                for (int off = 16; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
                const int row = m0 + tile_row<A_RC, 4>(wave, 2 * i + rr0);
                if ((lane & 31) == 0 && row < g.M) g.dot_out[row] = s;
            }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < WTM; ++i) {
#pragma unroll
        for (int j = 0; j < WTN; ++j) {
            int col = n0 + tile_row<B_RC, WTN>(j, lr);
            if (col >= g.N) continue;
            float bv = (!plain && g.bias) ? g.bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                int row = m0 + tile_row<A_RC, 4 * WTM>(wave * WTM + i, lq * 4 + r);
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

// ---- skinny GEMM: M <= 32 rows (the WGAN-GP step at the drivers' default batch of 8, train_gan.py:223-308) --------------
// The tiled kernel above gives such a product to a handful of workgroups that each walk all of K behind one barrier per
// 16 k (25-40 us for 8 x 800 x 800).  Here every workgroup owns 16 output columns, its eight waves split K between
// them, and the operands go straight from global memory into MFMA fragments (a lane's 16-byte load = its k values of
// four consecutive steps: k = 16*group + 4*quarter + step, the same free choice of k order as everywhere else) -- no
// LDS staging, no barrier in the loop; the partial tiles meet in LDS once at the end.
#define SK_WAVES 8   // waves per workgroup, K interleaved over them in 16-k groups
#define SK_UNROLL 4  // groups a wave has in flight: all their loads are issued before the first MFMA
template <int MT, bool BKC, bool VEC>
__global__ __launch_bounds__(64 * SK_WAVES) void gemm_skinny_kernel(GzslGemm g) {
    __shared__ float red[SK_WAVES][MT][256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 15, lq = lane >> 4;
    const int n0 = blockIdx.x * 16;
    const int n = n0 + lr;
    const bool n_ok = n < g.N;
    const float* bp = g.B + (long long)min(n, g.N - 1) * g.b_sn;
    const float* arow[MT];
    bool a_ok[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        const int m = 16 * t + lr;
        a_ok[t] = m < g.M;
        arow[t] = g.A + (long long)min(m, g.M - 1) * g.a_sm;
    }
    f32x4 acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    // split-K (long K, few columns): blockIdx.y owns k_per_split of K and stores its partial tile to slab blockIdx.y
    int kbeg = 0, kend = g.K;
    float* C = g.C;
    if (g.split_k > 1) {
        kbeg = blockIdx.y * g.k_per_split;
        kend = min(g.K, kbeg + g.k_per_split);
        C += (long long)blockIdx.y * g.slab_stride;
    }
    const int ngroups = (max(kend - kbeg, 0) + 15) / 16;
    for (int gi0 = wave; gi0 < ngroups; gi0 += SK_WAVES * SK_UNROLL) {
        float a[SK_UNROLL][MT][4], b[SK_UNROLL][4];
#pragma unroll
        for (int u = 0; u < SK_UNROLL; ++u) {
            const int k0 = kbeg + 16 * (gi0 + u * SK_WAVES) + 4 * lq;  // groups past the end read as zeros (k_ok false)
            if (VEC) {  // K, kbeg, kend % 4 == 0: the lane's four k are all inside the range or all outside
                const bool k_ok = k0 < kend;
                const int kc = k_ok ? k0 : 0;
#pragma unroll
                for (int t = 0; t < MT; ++t) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(arow[t] + kc);
                    const float keep = (a_ok[t] && k_ok) ? 1.f : 0.f;
#pragma unroll
                    for (int s = 0; s < 4; ++s) a[u][t][s] = v[s] * keep;
                }
                const float keepb = (n_ok && k_ok) ? 1.f : 0.f;
                if (BKC) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(bp + kc);
#pragma unroll
                    for (int s = 0; s < 4; ++s) b[u][s] = v[s] * keepb;
                } else {
#pragma unroll
                    for (int s = 0; s < 4; ++s) b[u][s] = bp[(long long)(kc + s) * g.b_sk] * keepb;
                }
            } else {
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const bool k_ok = k0 + s < kend;
                    const int kc = k_ok ? k0 + s : 0;
#pragma unroll
                    for (int t = 0; t < MT; ++t)
                        a[u][t][s] = arow[t][(long long)kc * g.a_sk] * ((a_ok[t] && k_ok) ? 1.f : 0.f);
                    b[u][s] = bp[(long long)kc * g.b_sk] * ((n_ok && k_ok) ? 1.f : 0.f);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < SK_UNROLL; ++u)
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int t = 0; t < MT; ++t) acc[t] = mfma16(a[u][t][s], b[u][s], acc[t]);
    }
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave][t][lane * 4 + r] = acc[t][r];
    __syncthreads();
    // thread tid < 256 finishes accumulator element (lane' = tid >> 2, r = tid & 3) of every row tile: row
    // 4*(lane' >> 4) + r, column lane' & 15; the waves' partial sums are added in a fixed order
    if (tid >= 256) return;
    const int ln = tid >> 2, r = tid & 3;
    const int col = n0 + (ln & 15);
    if (col >= g.N) return;
    const float bv = g.bias ? g.bias[col] : 0.f;
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        const int row = 16 * t + 4 * (ln >> 4) + r;
        if (row >= g.M) continue;
        float sum = 0.f;
#pragma unroll
        for (int w = 0; w < SK_WAVES; ++w) sum += red[w][t][tid];
        float v = g.alpha * sum + bv;
        float* cp = C + (long long)row * g.ldc + col;
        if (g.accumulate) v += *cp;
        if (g.act == GZSL_ACT_RELU) v = fmaxf(v, 0.f);
        else if (g.act == GZSL_ACT_TANH) v = tanhf(v);
        else if (g.act == GZSL_ACT_LRELU) v = v > 0.f ? v : 0.2f * v;
        *cp = v;
    }
}

static int launch_skinny(const GzslGemm& g, hipStream_t st) {
    const bool bkc = g.b_sk == 1;
    const bool vec = g.a_sk == 1 && (g.K & 3) == 0 && (g.a_sm & 3) == 0 && ((uintptr_t)g.A & 15) == 0 &&
                     (!bkc || ((g.b_sn & 3) == 0 && ((uintptr_t)g.B & 15) == 0));
    const dim3 grid(cdiv(g.N, 16), g.split_k > 1 ? g.split_k : 1), block(64 * SK_WAVES);
#define GZ_SK(MT, KC, V) hipLaunchKernelGGL((gemm_skinny_kernel<MT, KC, V>), grid, block, 0, st, g)
#define GZ_SK2(MT)                             \
    do {                                       \
        if (bkc && vec) GZ_SK(MT, true, true); \
        else if (vec) GZ_SK(MT, false, true);  \
        else if (bkc) GZ_SK(MT, true, false);  \
        else GZ_SK(MT, false, false);          \
    } while (0)
    if (g.M <= 16) GZ_SK2(1);
    else GZ_SK2(2);
#undef GZ_SK2
#undef GZ_SK
    GZSL_CHECK_LAUNCH();
    return 0;
}

// ---- short-K outer-product GEMM: the weight gradients of the WGAN-GP step at small batches (K = Bg .. 3 Bg <= 64) -----
// C[M,N] = alpha * A^T-stored(M,K) . B(K,N) (+ C), both operands contiguous along their rows (a_sm == 1, b_sn == 1).
// With K this short the MFMA tile kernel is all prologue and epilogue (7 us for 800 x 400 x 24); here a thread owns a
// 4 x 4 output block and reads its two float4 per k straight from global memory (L2-resident, 16 lanes share a row of B,
// 4 lanes a row of A), every load independent of the FMAs.
__global__ __launch_bounds__(256) void gemm_shortk_kernel(GzslGemm g) {
    const int tid = threadIdx.x, tn = tid & 15, tm = tid >> 4;
    const int m = blockIdx.x * 64 + 4 * tm, n = blockIdx.y * 64 + 4 * tn;
    if (m >= g.M || n >= g.N) return;
    const float* ap = g.A + m;
    const float* bp = g.B + n;
    f32x4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    int k = 0;
    for (; k + 8 <= g.K; k += 8) {
        f32x4 a[8], b[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            a[u] = *reinterpret_cast<const f32x4*>(ap + (long long)(k + u) * g.a_sk);
            b[u] = *reinterpret_cast<const f32x4*>(bp + (long long)(k + u) * g.b_sk);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] += a[u][i] * b[u];
    }
    for (; k < g.K; ++k) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(ap + (long long)k * g.a_sk);
        const f32x4 b = *reinterpret_cast<const f32x4*>(bp + (long long)k * g.b_sk);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] += a[i] * b;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f32x4* cp = reinterpret_cast<f32x4*>(g.C + (long long)(m + i) * g.ldc + n);
        f32x4 v = acc[i] * g.alpha;
        if (g.accumulate) v += *cp;
        *cp = v;
    }
}

static bool shortk_ok(const GzslGemm& g) {
    return g.K <= 64 && g.K >= 1 && g.split_k <= 1 && g.a_sm == 1 && g.b_sn == 1 && g.a_mb >= g.M && g.k_inner >= g.K &&
           g.a_shift == 0 && g.a_span >= 0x7FFFFFFFu && !g.bias && g.act == GZSL_ACT_NONE && (g.M & 3) == 0 &&
           (g.N & 3) == 0 && (g.a_sk & 3) == 0 && (g.b_sk & 3) == 0 && (g.ldc & 3) == 0 && ((uintptr_t)g.A & 15) == 0 &&
           ((uintptr_t)g.B & 15) == 0 && ((uintptr_t)g.C & 15) == 0 && g.M >= 64 && g.N >= 64;
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
        // fixed summation order (bitwise reproducible); eight loads in flight
        float s = 0.f;
        int zz = 0;
        for (; zz + 8 <= nsplit; zz += 8) {
            float t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = slab[(zz + u) * stride + i];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += t[u];
        }
        for (; zz < nsplit; ++zz) s += slab[zz * stride + i];
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

static bool mult4(long long v) { return (v & 3) == 0; }

// Can every 16-byte slot of this operand be one aligned vector load that is wholly inside or wholly outside the
// valid region?  rc: contiguous along rows (else along k).
static bool vec_ok(const void* p, bool rc, int rows, int K, int mb, long long mbs, int sm, int sk, int shift,
                   unsigned span, int k_inner, long long kbs) {
    if (((uintptr_t)p & 15) != 0) return false;
    const bool batched_rows = mb < rows, batched_k = k_inner < K;
    {   // the fast path keeps 32-bit element offsets: the operand's extent must stay below 2^31 floats
        const long long nb = batched_rows ? (rows + mb - 1) / mb : 1;
        const long long ext = nb * (mbs < 0 ? -mbs : mbs) + (long long)(batched_rows ? mb : rows) * (sm < 0 ? -sm : sm) +
                              (long long)K * (sk < 0 ? -sk : sk);
        if (ext >= (1ll << 31)) return false;
    }
    if (!mult4(shift) || (span != 0x7FFFFFFFu && span != 0xFFFFFFFFu && !mult4(span))) return false;
    if (batched_rows && !mult4(mbs)) return false;
    if (batched_k && !mult4(kbs)) return false;
    if (rc) {
        if (sm != 1 || !mult4(rows)) return false;
        if (K > 1 && !mult4(sk)) return false;
        if (batched_rows && !mult4(mb)) return false;
    } else {
        if (sk != 1 || !mult4(K)) return false;
        if (rows > 1 && !mult4(sm)) return false;
        if (batched_k && !mult4(k_inner)) return false;
    }
    return true;
}

template <int WTM, int WTN>
static int launch_cfg(const GzslGemm& g, hipStream_t st) {
    constexpr int BM = 64 * WTM;
    constexpr int BN = 16 * WTN;
    constexpr int BNP = BN;
    // an operand is staged row-contiguous only when it is NOT contiguous along k
    const bool arc = g.a_sk != 1 && g.a_sm == 1, brc = g.b_sk != 1 && g.b_sn == 1;
    const bool vec = vec_ok(g.A, arc, g.M, g.K, g.a_mb, g.a_mbs, g.a_sm, g.a_sk, g.a_shift, g.a_span, g.k_inner, g.a_kbs) &&
                     vec_ok(g.B, brc, g.N, g.K, 0x7FFFFFFF, 0, g.b_sn, g.b_sk, 0, 0xFFFFFFFFu, g.k_inner, g.b_kbs);
    size_t smem = 2 * sizeof(float) * ((arc ? BK * BM : BM * LDK) + (brc ? BK * BNP : BN * LDK) + 4 * NTHREADS);
    if (smem < sizeof(float) * 4 * 16 * 132) smem = sizeof(float) * 4 * 16 * 132;  // the epilogue's per-wave staging rows
    dim3 grid(cdiv(g.M, BM) * cdiv(g.N, BN), 1, g.split_k > 1 ? g.split_k : 1);
#define GZ_LAUNCH(AR, BR, V) \
    hipLaunchKernelGGL((gemm_f32_kernel<WTM, WTN, AR, BR, V>), grid, dim3(NTHREADS), smem, st, g)
#define GZ_PICK(V)                               \
    do {                                         \
        if (!arc && !brc) GZ_LAUNCH(false, false, V); \
        else if (!arc && brc) GZ_LAUNCH(false, true, V); \
        else if (arc && !brc) GZ_LAUNCH(true, false, V); \
        else GZ_LAUNCH(true, true, V);           \
    } while (0)
    if (vec) GZ_PICK(true);
    else GZ_PICK(false);
#undef GZ_PICK
#undef GZ_LAUNCH
    GZSL_CHECK_LAUNCH();
    return 0;
}

static int env_int(const char* name, int dflt) {
    const char* s = getenv(name);
    return s ? atoi(s) : dflt;
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
    if (g.dot_w) {  // fused head: one 64 x 208 tile row per workgroup, the vector epilogue (see the header)
        if (!g.dot_out || g.dot_period <= 0 || g.N > 208 || (g.N & 3) || (g.ldc & 3) || g.split_k > 1 || g.accumulate ||
            (((uintptr_t)g.C | (uintptr_t)g.bias | (uintptr_t)g.dot_w) & 15) != 0)
            return GZSL_EINVAL;
        return launch_cfg<1, 13>(g, st);
    }
    // few rows, plain operands: the skinny kernel (columns over workgroups, K over the waves of a workgroup)
    static const int no_skinny = env_int("GZSL_GEMM_NO_SKINNY", 0);
    if (!no_skinny && g.M <= 32 && g.K >= 32 && g.a_mb >= g.M && g.k_inner >= g.K && g.a_shift == 0 &&
        g.a_span >= 0x7FFFFFFFu)
        return launch_skinny(g, st);
    if (!no_skinny && shortk_ok(g)) {  // K <= 64 outer products (GZSL_GEMM_NO_SKINNY=1 disables both special kernels)
        hipLaunchKernelGGL(gemm_shortk_kernel, dim3(cdiv(g.M, 64), cdiv(g.N, 64)), dim3(256), 0, st, g);
        GZSL_CHECK_LAUNCH();
        return 0;
    }
    // tile width: the one whose padded N wastes least MFMA work (ties -> wider tile) ...
    const int pad128 = cdiv(g.N, 128) * 128, pad208 = cdiv(g.N, 208) * 208, pad112 = cdiv(g.N, 112) * 112;
    int wn = (pad208 < pad128 && pad208 <= pad112) ? 13 : (pad112 < pad128 ? 7 : 8);
    // ... unless the grid is too shallow to reach every CU twice (GAN MLPs at Bg = 4096, small label graphs): then a cost
    // model fitted to tools/sweep_gemm_wn.py picks the width.  A CU's time is (workgroups it receives) x (tile width)
    // over a throughput factor for 1 / 2 / >= 3 co-resident workgroups; narrower tiles win up to 1.5x there.  Deep
    // and split-K grids keep the padding rule (inside the train step the narrow tiles measured slower).
    const long rows64 = cdiv(g.M, 64);
    if (g.split_k <= 1 && rows64 * cdiv(g.N, 16 * wn) < 512) {
        float best = 0.f;
        const int cand[3] = {13, 8, 7};
        for (int c = 0; c < 3; ++c) {
            const long per_cu = (rows64 * cdiv(g.N, 16 * cand[c]) + 255) / 256;
            const float thr = per_cu >= 3 ? 1.0f : (per_cu == 2 ? 0.95f : 0.88f);
            const float cost = (float)per_cu * cand[c] / thr;
            if (c == 0 || cost < best * 0.995f) best = cost, wn = cand[c];
        }
    }
    static const int force_wn = env_int("GZSL_GEMM_WN", 0);  // diagnostic override: 13, 7 or 8
    if (force_wn == 13 || force_wn == 7 || force_wn == 8) wn = force_wn;
    // tile height: 128 rows once the grid is deep enough that the last round does not matter, else 64
    static const int force_wtm = env_int("GZSL_GEMM_WTM", 0);
    const long tiles64 = (long)cdiv(g.M, 64) * cdiv(g.N, 16 * wn) * (g.split_k > 1 ? g.split_k : 1);
    int wtm = tiles64 >= 8192 ? 2 : 1;
    if (force_wtm == 1 || force_wtm == 2) wtm = force_wtm;
    if (wtm == 2) {
        if (wn == 13) return launch_cfg<2, 13>(g, st);
        if (wn == 7) return launch_cfg<2, 7>(g, st);
        return launch_cfg<2, 8>(g, st);
    }
    if (wn == 13) return launch_cfg<1, 13>(g, st);
    if (wn == 7) return launch_cfg<1, 7>(g, st);
    return launch_cfg<1, 8>(g, st);
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
