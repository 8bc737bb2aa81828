/* gzsl_hip.h - C ABI of libgzsl_hip.so: MI355X (gfx950) kernels for the gzsl_text hot path.
 *
 * The reference (petuum/gzsl_text) has no FFI of its own: its hot path is Python nn.Modules whose ops dispatch to
 * ATen/cuDNN/cuBLAS.  Each entry point below names the reference statement(s) it replaces (path:line under the
 * reference tree).  All pointers are DEVICE pointers to contiguous row-major buffers (fp32 unless said otherwise,
 * token / label ids int64 as the reference's LongTensors); `stream` is a hipStream_t; nothing allocates or syncs.
 * Return value: 0 on success, a hipError_t (>0) from the launch, GZSL_EINVAL (1001) for unsupported arguments, or
 * GZSL_ETOOBIG (1002) where an entry point states a size limit.
 * The host side (gzsl_text_amd/, Python, mirroring the modules package) binds these with ctypes; see INTEGRATION.md.
 */
#ifndef GZSL_HIP_H
#define GZSL_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GZSL_ACT_NONE 0
#define GZSL_ACT_RELU 1
#define GZSL_ACT_TANH 2
#define GZSL_ACT_LRELU 3 /* LeakyReLU(0.2), generative_module.py:46,61 */

/* ---- dense contractions ---------------------------------------------------------------------------------------
 * C[M,N] = act(alpha * A(M,K).B(K,N) + bias[n] + (accumulate ? C : 0)), exact-fp32 MFMA.
 * Replaces nn.Conv1d (icd_modules.py:74,131 - implicit GEMM over k*E-float windows), nn.Linear
 * (icd_modules.py:68,76,134,159; generative_module.py:44-45,59-60; decoder_module.py:29,40,43), nn.GRUCell's two
 * GEMMs (graph_module.py:87,122) and the autograd products of all of them. */
typedef struct GzslGemm {
    const float* A;
    const float* B;
    float* C;
    const float* bias; /* [N] or NULL */
    int M, N, K;
    int a_mb;          /* m -> (m / a_mb, m % a_mb); >= M when A is not batched along m */
    long long a_mbs;   /* elements between A batches */
    int a_sm, a_sk;    /* element strides of A inside a batch */
    int a_shift;       /* added to the in-batch offset (may be negative: conv data-gradient window start) */
    unsigned a_span;   /* A element reads as 0 unless 0 <= in-batch offset < a_span */
    int k_inner;       /* k -> (k / k_inner, k % k_inner); >= K when k is not batched; otherwise >= 32 */
    long long a_kbs;   /* elements between A k-batches */
    int b_sk, b_sn;    /* element strides of B */
    long long b_kbs;   /* elements between B k-batches */
    int ldc;
    int act;
    int accumulate;
    float alpha;
    int split_k;           /* > 1: blockIdx.z owns k_per_split of K and writes alpha*partial to C + z*slab_stride */
    int k_per_split;       /* multiple of 32 */
    long long slab_stride; /* elements */
    /* Optional fused classifier head (icd_modules.py:159 + 177-182): with dot_w set, every output row
     * m ALSO yields dot_out[m] = sum_n relu(C[m,n]) * dot_w[(m % dot_period) * N + n] from the tile in registers
     * (C = the epilogue's value: alpha*A.B + bias, act applied), i.e. logits[b,l] = <relu(direct[b,l,:]), clf[l,:]>
     * without a second pass over `direct`; dot_no_store = 1 additionally skips writing C (forward-only scoring never
     * reads it).  Needs one tile across N (N <= 208), N % 4 == 0, ldc % 4 == 0, 16-byte aligned C / bias / dot_w,
     * no split-K, no accumulate (else GZSL_EINVAL). */
    const float* dot_w;
    float* dot_out;
    int dot_period;
    int dot_no_store;
} GzslGemm;
int gzsl_gemm(const GzslGemm* g, void* stream);
/* out = act(sum_z slab[z] + bias) (+ out); unpack_k > 0 scatters a [(j*E+e), f] slab into conv weight layout [F,E,k] */
int gzsl_splitk_reduce(const float* slab, long long slab_stride, int nsplit, float* out, int M, int N, int ldo,
                       const float* bias, int act, int accumulate, int unpack_k, int unpack_e, void* stream);

/* ---- fused label attention ------------------------------------------------------------------------------------
 * icd_modules.py:137-142: scores, softmax over tokens, attn dropout(0.2), pooling - never materialising B x L x T'.
 * PX [B,Tp,E] keys, X [B,Tp,F] values, Q [L,E] queries (this rank's label rows), O [B,L,F], LSE [B,L].
 * Dropout: counter hash on (b, l_offset + l, t) with L_total as the row pitch (oracle/rng.py); drop_thresh = 0 disables. */
int gzsl_label_attn_fwd(const float* PX, const float* X, const float* Q, float* O, float* LSE, int B, int Tp, int E,
                        int F, int L, int l_offset, int L_total, uint32_t seed, uint32_t drop_thresh, float drop_scale,
                        void* stream);
/* EXPERIMENTAL, opt-in (never the default path): the same forward with every fp32 product formed on the bf16 matrix pipe
 * from three-way bf16 splits of both operands (six cross terms, fp32 accumulation; csrc/attn_bf16.hip states the error
 * analysis, tools/bf16_split_gemm.hip measures it).  Same arguments and results (same tolerances in the tests) as
 * gzsl_label_attn_fwd; E <= 256, F <= 208; `work` = 16-byte aligned device scratch of at least
 * gzsl_label_attn_fwd_bf16x6_work_bytes(B, Tp, E, F) bytes (the repacked key / value tiles: 80 MB at cfg2). */
long long gzsl_label_attn_fwd_bf16x6_work_bytes(int B, int Tp, int E, int F);
int gzsl_label_attn_fwd_bf16x6(const float* PX, const float* X, const float* Q, float* O, float* LSE, int B, int Tp, int E,
                               int F, int L, int l_offset, int L_total, uint32_t seed, uint32_t drop_thresh,
                               float drop_scale, void* work, long long work_bytes, void* stream);
/* delta[row] = sum_f dO[row,f] * O[row,f] (softmax-backward row constant), rows = B*L */
int gzsl_attn_delta(const float* dO, const float* O, float* delta, long long rows, int F, void* stream);
/* autograd of the above: dPX [B,Tp,E], dX [B,Tp,F], dQ [L,E] are ACCUMULATED with float atomics (caller zeroes or
 * pre-loads them).  nsplit label splits per (note, 128-token tile) raise the workgroup count past 256 CUs. */
int gzsl_label_attn_bwd(const float* dO, const float* PX, const float* X, const float* Q, const float* LSE,
                        const float* DELTA, float* dPX, float* dX, float* dQ, int B, int Tp, int E, int F, int L,
                        int l_offset, int L_total, int nsplit, uint32_t seed, uint32_t drop_thresh, float drop_scale,
                        void* stream);
/* Opt-in, beside gzsl_label_attn_fwd_bf16x6: the same backward (autograd of icd_modules.py:137-142; same arguments,
 * accumulation contract and test tolerances as gzsl_label_attn_bwd) with every fp32 product formed as six bf16 MFMAs
 * (csrc/attn_bwd_bf16.hip).  E <= 112, F <= 208 (GZSL_EINVAL beyond: callers fall back to gzsl_label_attn_bwd); 64-token
 * workgroups; `work` = 16-byte aligned device scratch of at least gzsl_label_attn_bwd_bf16x6_work_bytes(B, Tp, E, F, L)
 * bytes (the bf16 plane images of Q and dO per 32-label tile, 218 MB at cfg2, and one slab of partial dQ tiles per
 * (note, 64-token tile), 2.5 GB at cfg2, summed in a fixed order by a second launch). */
long long gzsl_label_attn_bwd_bf16x6_work_bytes(int B, int Tp, int E, int F, int L);
int gzsl_label_attn_bwd_bf16x6(const float* dO, const float* PX, const float* X, const float* Q, const float* LSE,
                               const float* DELTA, float* dPX, float* dX, float* dQ, int B, int Tp, int E, int F, int L,
                               int l_offset, int L_total, int nsplit, uint32_t seed, uint32_t drop_thresh,
                               float drop_scale, void* work, long long work_bytes, int defer_dq_reduce, void* stream);
/* defer_dq_reduce != 0 leaves dPX / dX complete and dQ untouched: the sum of the dQ slabs in `work` is then added by this
 * call (same B, Tp, E, F, L), on any stream ordered after the first -- e.g. beside the encoder's backward, which does
 * not need dQ (train_base.py:112's autograd has no such split; it exists for stream overlap only). */
int gzsl_label_attn_bwd_bf16x6_dq_reduce(const void* work, float* dQ, int B, int Tp, int E, int F, int L, void* stream);

/* Deterministic form of gzsl_label_attn_bwd (SURVEY 5 asks for a deterministic-reduction mode; the reference itself is
 * only as reproducible as torch's atomics-based embedding / index_add backward): the partial sums leave as plain stores
 * into `work` -- one [L,E] slab of dQ per (note, token tile), one [B,Tp,E] / [B,Tp,F] slab of dPX / dX per label split --
 * and are then ADDED to dPX / dX / dQ in a fixed order.  Same arguments and result (up to summation order) as the
 * atomic entry point; two calls on the same inputs give bit-identical outputs.  work_floats must be at least
 * gzsl_label_attn_bwd_det_work_floats(...) (1.3 GB at cfg2: a mode, not the fast path). */
long long gzsl_label_attn_bwd_det_work_floats(int B, int Tp, int E, int F, int L, int nsplit);
int gzsl_label_attn_bwd_det(const float* dO, const float* PX, const float* X, const float* Q, const float* LSE,
                            const float* DELTA, float* dPX, float* dX, float* dQ, int B, int Tp, int E, int F, int L,
                            int l_offset, int L_total, int nsplit, uint32_t seed, uint32_t drop_thresh, float drop_scale,
                            float* work, long long work_floats, void* stream);
/* Library-wide switch for the small reductions that otherwise end in a handful of float atomics (gzsl_ldam_bce's loss:
 * one block; gzsl_colsum_ws: always through the workspace).  Returns the previous setting.  Not a stream operation. */
int gzsl_set_deterministic(int on);

/* ---- embedding / label side -------------------------------------------------------------------------------------
 * out[row,:] = W[ids[row],:] * mask[row] * dropout;  common_module.py:87-88 + icd_modules.py:153-157 (mask may be NULL).
 * row_offset = global index of row 0 (b0*T when a rank encodes notes b0.. of the batch): the dropout hash is keyed on it. */
int gzsl_emb_gather_fwd(const float* W, const int64_t* ids, const float* mask, float* out, long long rows, int E,
                        long long row_offset, uint32_t seed, uint32_t drop_thresh, float drop_scale, void* stream);
/* dW[ids[row],:] += dOut[row,:] * mask[row] * dropout, skipping ids == pad_idx (nn.Embedding padding_idx, common_module.py:76) */
int gzsl_emb_scatter_add(float* dW, const int64_t* ids, const float* mask, const float* dOut, long long rows, int E,
                         int64_t pad_idx, long long row_offset, uint32_t seed, uint32_t drop_thresh, float drop_scale,
                         void* stream);
/* Same result (up to fp32 summation order), for id streams with heavy repeats (word ids are Zipf-distributed): each
 * workgroup sorts the ids of 128 consecutive rows in LDS and sums runs of equal ids before the atomics.  E <= 297. */
int gzsl_emb_scatter_add_grouped(float* dW, const int64_t* ids, const float* mask, const float* dOut, long long rows,
                                 int E, int64_t pad_idx, long long row_offset, uint32_t seed, uint32_t drop_thresh,
                                 float drop_scale, void* stream);
/* icd_modules.py:110-113 average_label_embedding: out[n,:] = sum_j mask[n,j] W[idx[n,j],:] / sum_j mask[n,j] */
int gzsl_label_avg_fwd(const float* W, const int64_t* idx, const float* mask, float* out, int N, int Td, int E,
                       void* stream);
int gzsl_label_avg_bwd(float* dW, const float* davg, const int64_t* idx, const float* mask, int N, int Td, int E,
                       int64_t pad_idx, void* stream);
/* graph_module.py:114-117 (and :70-73): out[n,:] (+)= post[n] * sum_p val[p] * pre[col[p]] * x[col[p],:]  over CSR row n;
 * pre/post may be NULL (=1).  post = 1/(deg+1e-7) gives the neighbour mean; pre with the transposed CSR gives its adjoint.
 * accumulate: 0 store, 1 out += (plain), 2 out += with float atomics, empty rows skipped -- the adjoint of
 * average_label_embedding over the inverted description index (word -> labels), which adds into the word-table gradient
 * beside the note encoder's scatter-add: one atomic per (word, column) instead of one per (label, slot, column). */
int gzsl_spmm_csr(const int64_t* rowptr, const int64_t* col, const float* val, const float* x, const float* pre,
                  const float* post, float* out, int N, int H, int accumulate, void* stream);
/* nn.GRUCell gate math (graph_module.py:122) after the two GEMMs gi = u.Wih^T + bih, gh = h.Whh^T + bhh ([N,3H], r|z|n) */
int gzsl_gru_gate_fwd(const float* gi, const float* gh, const float* h, float* hn, int N, int H, void* stream);
int gzsl_gru_gate_bwd(const float* dhn, const float* gi, const float* gh, const float* h, float* dgi, float* dgh,
                      float* dh, int N, int H, void* stream);
/* One step t of the packed label GRU, inference (lm_module.py:68-88 with training=False; pack_padded_sequence /
 * pad_packed_sequence semantics): rows with sum(mask[b,:]) > t take the GRUCell update of h in place, others keep
 * their state.  gi row b = gi + b*gi_stride holds x_t.W_ih^T + b_ih, gh [Bg,3H] = h.W_hh^T + b_hh.  pool_mode 0: none
 * ('last': the answer is h); 1: pool = max(pool, h_t) over non-zero h_t (caller initialises pool to -inf);
 * 2: pool += h_t, divided by the length at t == T-1 (caller initialises to 0). */
int gzsl_gru_seq_step(const float* gi, long long gi_stride, const float* gh, float* h, const float* mask, int T, int t,
                      float* pool, int pool_mode, int Bg, int H, void* stream);

/* ---- classifier head + loss -------------------------------------------------------------------------------------
 * icd_modules.py:177-182 (and :263-272): logits[b,l] = sum_d clf[l,d] * relu(direct[b,l,d]) */
int gzsl_logits_fwd(const float* direct, const float* clf, float* logits, int B, int L, int D, void* stream);
/* ddirect[b,l,d] = dlogits[b,l] * clf[l,d] * (direct > 0);  dclf[l,d] (+)= sum_b dlogits[b,l] * relu(direct[b,l,d]) */
int gzsl_logits_bwd(const float* direct, const float* clf, const float* dlogits, float* ddirect, float* dclf, int B,
                    int L, int D, int accumulate_dclf, void* stream);
/* The same two gradients (dclf written, not accumulated) plus delta[b,l] = sum_f dO[b,l,f]*O[b,l,f], the row constant
 * gzsl_label_attn_bwd needs, obtained WITHOUT dO or O: dO = ddirect.Wo and direct = O.Wo^T + bo (icd_modules.py:159) give
 * delta = sum_d ddirect[b,l,d]*(direct[b,l,d] - bo[d]); replaces gzsl_attn_delta's pass over 2*B*L*F floats.
 * D % 4 == 0, D <= 1024, 16-byte aligned buffers (else GZSL_EINVAL: use gzsl_logits_bwd + gzsl_attn_delta). */
int gzsl_logits_bwd_delta(const float* direct, const float* clf, const float* bo, const float* dlogits, float* ddirect,
                          float* dclf, float* delta, int B, int L, int D, void* stream);
/* icd_modules.py:162-172 + BCEWithLogitsLoss(sum) common_module.py:29-30: z = logit - y*margin[l]*C;
 * *loss += sum bce(z,y) * inv_b (atomic; caller zeroes); dlogits = (sigmoid(z) - y) * inv_b.  margin may be NULL. */
int gzsl_ldam_bce(const float* logits, const float* y, const float* margin, float C, float inv_b, float* loss,
                  float* dlogits, float* probs, int B, int L, void* stream);
/* icd_modules.py:233-249 forward_final: z_i = <fc[lab_i,:], feat_i> - margin[lab_i]*C; *loss += sum bce(z,1)*scale;
 * dfc[lab_i,:] += (sigmoid(z_i)-1)*scale*feat_i (atomic); dfeat may be NULL */
int gzsl_forward_final(const float* fc, const float* feats, const int64_t* labels, const float* margin, float C,
                       float scale, float* loss, float* dfc, float* dfeat, int Bs, int D, void* stream);
/* common_module.py:138-182 AdamW.step (no bias correction, eps added to sqrt(v), wd on the parameter) */
int gzsl_adamw_nobias_step(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1,
                           float beta2, float eps, float wd, void* stream);

/* ---- small elementwise / reductions used by the backward passes ------------------------------------------------ */
int gzsl_act_bwd(float* d, const float* y, long long n, int act, void* stream); /* d *= act'(y), y = activated output */
int gzsl_colsum(const float* x, float* out, long long M, int N, int ld, void* stream);
/* Same, with a device workspace of work_floats floats (contents undefined): row blocks write partial sums there and a
 * second launch adds them to out in a fixed order, instead of up to 4096 same-address atomics per column. */
int gzsl_colsum_ws(const float* x, float* out, long long M, int N, int ld, float* work, long long work_floats,
                   void* stream); /* out[n] += sum_m x[m*ld+n] */
/* Allocation-free variant of gzsl_colsum_ws.  No counterpart in the reference: it is for callers that cannot allocate
 * device memory (code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc) and therefore
 * cannot bring a workspace.  The partial sums live in a 2 MiB file-scope device array inside the library (one copy per
 * GPU, held until the process ends; its address is looked up with hipGetSymbolAddress at the first call on each
 * device, which must not itself be under stream capture).  Calls no hipMalloc / hipFree.  N > GZSL_COLSUM_STATIC_MAX_N
 * is rejected with GZSL_ETOOBIG.  All calls on one GPU share the array: two calls that overlap on the same GPU (other
 * streams or threads) race; nothing needs clearing between calls (every word read is written first by the same call). */
#define GZSL_COLSUM_STATIC_MAX_N 8192
#define GZSL_ETOOBIG 1002
int gzsl_colsum_static(const float* x, float* out, long long M, int N, int ld, void* stream); /* out[n] += sum_m x[m*ld+n] */
/* conv_modules.0.weight [F,E,k] -> Wp[f][j*E+e] (forward GEMM operand) and Wflip[e][j'*F+f] = W[f][e][k-1-j'] (data grad) */
int gzsl_conv_weight_pack(const float* W, float* Wp, float* Wflip, int F, int E, int k, void* stream);

/* ---- batching and eval scoring either side of the step ---------------------------------------
 * utils/helper.py:117-123 iterate_minibatch: y = zeros[B,L]; y[row, c_idx] = 1.  y_indices [B,W] int64 is the ragged
 * label list of the batch padded with -1 (the reference's own y_indices, cut to the widest row); entries outside
 * [0,L) are skipped -- the host mirror raises IndexError for them first, as numpy does. */
int gzsl_multihot(const int64_t* y_indices, int W, float* y, int B, int L, void* stream);
/* train_base.py:133 + utils/metrics.py:29,63-121: p = sigmoid(logits[b,l]) (written to probs [B,ld_probs] unless NULL),
 * yhat = round(p) with numpy's half-to-even (p == 0.5 -> 0); counts [4,Lc] int32 (caller zeroes before the first batch):
 * [0] += yhat&y, [1] += yhat, [2] += (y != 0), [3] += yhat|y over the B rows.  Every macro/micro accuracy, precision,
 * recall and F1 of metrics.py, for any label subset, follows from these integers; the AUCs need the scores on the host. */
int gzsl_eval_counts(const float* logits, int ld_logits, const float* y, int ld_y, int* counts, float* probs,
                     int ld_probs, int B, int Lc, void* stream);

/* ---- WGAN-GP step ---------------------------------------------------------------------------------------------
 * generative_module.py:27 interpolates = alpha*real + (1-alpha)*fake, alpha [Bg] */
int gzsl_gp_interp(const float* real, const float* fake, const float* alpha, float* out, int Bg, int D, void* stream);
/* closed-form gradient penalty pieces (generative_module.py:22-38): v[b,j] = slope(h[b,j]) * w2[j], slope in {1,0.2} */
int gzsl_gp_slope_v(const float* h, const float* w2, float* v, float* s, int Bg, int H, void* stream);
/* nrm_b = ||g_b||; *gp += lam*(nrm_b-1)^2/Bg (atomic); g_b <- (2*lam*(nrm_b-1)/(Bg*nrm_b)) * g_b  (in place) */
int gzsl_gp_rownorm(float* g, float* gp, int Bg, int D, float lam, void* stream);
/* dw2[j] += sum_b dv[b,j]*s[b,j] */
int gzsl_gp_dw2(const float* dv, const float* s, float* dw2, int Bg, int H, void* stream);
/* out[b] = dot(act(a[b,:]), w) + bias0 (critic fc2 with one output, generative_module.py:60,67-68); bias may be NULL.
 * lrelu_in != 0: `a` holds pre-activations and LeakyReLU(0.2) is applied on the fly. */
int gzsl_rowdot(const float* a, const float* w, const float* bias, float* out, int Bg, int H, int lrelu_in,
                void* stream);
/* *out += sum_b x[b] * w[b] * scale (critic means, train_gan.py:244,252,300); w may be NULL */
int gzsl_weighted_sum(const float* x, const float* w, float scale, float* out, int n, void* stream);
/* da[b,j] = dout[b] * w[j] * slope(act[b,j]);  dw[j] += sum_b dout[b]*act[b,j];  db += sum_b dout[b]  (fc2 backward, one output) */
int gzsl_rowdot_bwd(const float* dout, const float* w, const float* act, float* da, float* dw, float* db, int Bg,
                    int H, int lrelu_in, void* stream);
/* decoder_module.py:39-59 soft-target cross-entropy over the keyword vocabulary: loss_b = -sum_v log_softmax(logits)[b,v]*t[b,v],
 * t = normalised mask scattered by assignment (last duplicate wins).  Writes loss_b and, in place, dlogits = w_b*(softmax*sum(t) - t). */
int gzsl_keyword_ce(float* logits, const int64_t* kw_idx, const float* kw_mask, const float* w, float* loss_b, int Bg,
                    int Vk, int top_k, void* stream);

/* Same with the driver's row weighting folded in (train_gan.py:288-296): keywords_weight_b = (sim == NULL || sim[b] == 1)
 * ? 1 : 0; dlogits rows are scaled by keywords_weight_b * grad_scale and *loss_sum += loss_b * keywords_weight_b *
 * loss_scale (atomic; loss_sum may be NULL). */
int gzsl_keyword_ce_sim(float* logits, const int64_t* kw_idx, const float* kw_mask, const float* sim, float grad_scale,
                        float loss_scale, float* loss_b, float* loss_sum, int Bg, int Vk, int top_k, void* stream);
/* One critic iteration's elementwise middle in one launch (train_gan.py:244-256, generative_module.py:22-38, 59-68):
 * between the fc1 GEMMs and the gradient GEMMs.  hf [2Bg,H] = [real; fake] @ fc1.weight[:, :D]^T, lh [Bg,H] = lab @
 * fc1.weight[:, D:]^T + fc1.bias; pre-activations h = hf + lh; wb = sim[b]/Bg (sim NULL: 1/Bg); slope s in {1, 0.2}.
 *   lst [3Bg,H]: rows [0,Bg) = -wb*w2*s(h_real), [Bg,2Bg) = +wb*w2*s(h_fake), [2Bg,3Bg) = s(h_hat)*w2 with
 *   h_hat = alpha*h_real + (1-alpha)*h_fake;  dhs [Bg,H] = the first two row blocks added;  sl [Bg,H] = s(h_hat);
 *   dw2[j] += sum_b wb*(lrelu(h_fake) - lrelu(h_real))[b,j];  db1[j] += sum_b dhs[b,j] (d fc1.bias);
 *   *cost += critic_fake - critic_real   (all three through float atomics).
 * d fc2.bias of the critic objective is identically 0 and is not written. */
int gzsl_gan_critic_head(const float* hf, const float* lh, const float* w2, const float* sim, const float* alpha,
                         float* lst, float* dhs, float* sl, float* dw2, float* db1, float* cost, int Bg, int H,
                         void* stream);
/* Generator iteration (train_gan.py:299-306), critic frozen: dh[b,j] = -wb*w2[j]*s(h[b,j]), h = hf + lh [Bg,H];
 * *cost += -sum_b wb*(lrelu(h_b) . w2 + b2[0]) (atomic). */
int gzsl_gan_gen_head(const float* hf, const float* lh, const float* w2, const float* b2, const float* sim, float* dh,
                      float* cost, int Bg, int H, void* stream);
/* torch.optim.Adam.step (train_gan.py:155-157: betas (0.5, 0.999), no weight decay, no amsgrad) over ONE flat buffer of
 * n parameters (16-byte aligned): m += (g-m)(1-beta1); v = beta2 v + (1-beta2) g^2;
 * p -= lr/(1-beta1^t) * m / (sqrt(v)/sqrt(1-beta2^t) + eps).  state: GZSL_ADAM_STATE_INTS int32, zero before the first
 * step and owned by (p, n) from then on: every workgroup keeps its own copy of the step count there (all equal between
 * calls; state[0] is the number of steps taken).  The count lives on the device so that a recorded HIP graph replays
 * with the right bias correction; private copies mean no atomics and no ordering between workgroups. */
#define GZSL_ADAM_STATE_INTS 1024
int gzsl_adam_flat_step(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2,
                        float eps, int* state, void* stream);

/* library self-description: number of exported compute entry points and build target string */
const char* gzsl_build_info(void);

#ifdef __cplusplus
}
#endif
#endif /* GZSL_HIP_H */
