"""Tensor-level operations of the hot path, each one or two launches of libgzsl_hip.so.

`HipOps` is the only kernel provider the product uses.  The engine (engine.py, gan_engine.py,
parallel.py) is written against this interface so that its orchestration -- not its arithmetic --
can also be exercised on CPU by the test suite's stand-in (tests/ops_cpu.py).  Nothing here falls
back to torch arithmetic: torch supplies device memory (empty/zeros), views and streams only.
"""
import os

import torch

from . import hip
from .hip import ACT_NONE, ACT_RELU, ACT_TANH, ACT_LRELU, GzslGemm, call, dptr, ptr  # noqa: F401

INT_MAX = 2 ** 31 - 1


def drop_params(p):
    """(threshold, keep scale) of the counter RNG for drop probability p (threshold computed in double)."""
    if p is None or p <= 0.0:
        return 0, 1.0
    return min(int(float(p) * 4294967296.0), 0xFFFFFFFF), 1.0 / (1.0 - float(p))


def _check_f32(*ts):
    for t in ts:
        if t is not None and t.dtype != torch.float32:
            raise RuntimeError(f'expected float32 tensor, got {t.dtype}')


_DETERMINISTIC = False


def set_deterministic(on):
    """The deterministic-reduction mode is PROCESS-WIDE: it sets the library's switch (gzsl_set_deterministic: one-block
    loss sum, workspace column sums) and serialises the second stream (streams.Beside), and every HipOps of the process
    then takes the slab / sorted-gather forms of the attention and embedding gradients.  Returns the previous setting."""
    global _DETERMINISTIC
    from .streams import Beside
    prev = _DETERMINISTIC
    _DETERMINISTIC = bool(on)
    Beside.serial = _DETERMINISTIC
    hip.load().gzsl_set_deterministic(1 if _DETERMINISTIC else 0)
    return prev


def det_attn_bwd_work_bytes(B, Tp, E, F, L, nsplit):
    """Workspace of the deterministic attention backward (gzsl_label_attn_bwd_det): one [L,E] dQ slab per (note, token
    tile of 128 (E <= 100) or 64 tokens) and one [B,Tp,E+F] slab per label split."""
    ntt = -(-Tp // (128 if E <= 100 else 64))
    return 4 * (B * ntt * L * E + nsplit * B * Tp * (E + F))


class HipOps:
    """Deterministic mode (`HipOps(deterministic=True)`, GZSL_DETERMINISTIC=1, or ops_hip.set_deterministic(True)):
    every cross-workgroup sum of the train step takes a fixed-order form -- attention gradients through slabs + ordered
    reduce (gzsl_label_attn_bwd_det), the embedding gradient as a gather over a sorted index instead of float atomics,
    loss / column sums without atomics, one stream.  Two runs on the same inputs are then bit-identical
    (tests/test_gpu_model.py); it costs memory (det_attn_bwd_work_bytes: 1.3 GB at cfg2) and ~60 % of the step.
    The mode is process-wide (see set_deterministic): `deterministic=True` switches it on for every provider,
    an explicit `deterministic=False` switches it off, and the default (None) follows the environment / current mode."""
    name = 'hip'

    def __init__(self, deterministic=None):
        import os
        if deterministic is None:
            if os.environ.get('GZSL_DETERMINISTIC') == '1' and not _DETERMINISTIC:
                set_deterministic(True)
        else:
            set_deterministic(bool(deterministic))
        self.attn_bf16x6 = os.environ.get('GZSL_ATTN_BF16X6') == '1'  # opt-in bf16-pipe attention kernels, off by default
        self.defer_dq = False      # set by the engine around its attention backward (bf16x6 path only)
        self._pending_dq = []

    @property
    def deterministic(self):
        return _DETERMINISTIC

    # ------------------------------------------------------------------------------------ GEMM plumbing
    @staticmethod
    def _split_policy(M, N, K):
        """Split K when the output has too few 64-row tiles to fill 256 CUs x 3 workgroups (weight gradients: small
        MxN, huge K).  Every split keeps at least 256 k (16 k-tiles) so that the prologue / epilogue stay amortised."""
        if M <= 32:  # skinny kernel: 16 columns per workgroup; split a long K until ~256 workgroups, >= 512 k each
            split = min(256 // ((N + 15) // 16), K // 512)
            if split <= 1:
                return 1, 0
            kper = ((K + split - 1) // split + 31) // 32 * 32
            return (K + kper - 1) // kper, kper
        bn = min((208, 112, 128), key=lambda w: ((N + w - 1) // w * w, -w))
        tiles = ((M + 63) // 64) * ((N + bn - 1) // bn)
        if tiles >= 384 or K < 1024:
            return 1, 0
        split = min((768 + tiles - 1) // tiles, K // 256)
        if split <= 1:
            return 1, 0
        kper = ((K + split - 1) // split + 31) // 32 * 32
        split = (K + kper - 1) // kper
        return split, kper

    def _launch(self, g, M, N, K, out, bias, act, accumulate, split, kper, unpack=None):
        g.M, g.N, g.K = M, N, K
        g.alpha = g.alpha if g.alpha else 1.0
        if split > 1 or unpack is not None:
            if split <= 1:
                # one real slab: blockIdx.z = 1 gets an empty K range and stores zeros, the reduce kernel then
                # applies the unpack / accumulate / epilogue exactly as for a real split
                split, kper = 2, max(32, (K + 31) // 32 * 32)
            slab = torch.empty((split, M, N), dtype=torch.float32, device=out.device)
            g.C, g.ldc, g.bias, g.act, g.accumulate = dptr(slab), N, None, ACT_NONE, 0
            g.split_k, g.k_per_split, g.slab_stride = split, kper, M * N
            call('gzsl_gemm', g)
            uk, ue = unpack if unpack is not None else (0, 0)
            ldo = out.stride(0) if (unpack is None and out.dim() == 2) else N
            call('gzsl_splitk_reduce', dptr(slab), M * N, split, dptr(out), M, N, ldo,
                 ptr(bias), act, int(accumulate), uk, ue)
        else:
            g.C, g.ldc = dptr(out), out.stride(0)
            g.bias, g.act, g.accumulate = ptr(bias), act, int(accumulate)
            g.split_k, g.k_per_split, g.slab_stride = 1, 0, 0
            call('gzsl_gemm', g)
        return out

    def mm(self, A, B, out=None, bias=None, act=ACT_NONE, accumulate=False, alpha=1.0, allow_split=True):
        """out[M,N] = act(alpha * A @ B + bias (+ out)); A, B arbitrary-strided 2-D views, out row-major rows."""
        _check_f32(A, B, out, bias)
        M, K = A.shape
        K2, N = B.shape
        assert K == K2, (A.shape, B.shape)
        if out is None:
            assert not accumulate
            out = torch.empty((M, N), dtype=torch.float32, device=A.device)
        assert out.shape == (M, N) and out.stride(1) == 1
        if not A.is_cuda:
            raise RuntimeError('gzsl_text_amd kernels need tensors on a HIP device; there is no CPU path')
        g = GzslGemm()
        g.A, g.B = dptr(A), dptr(B)
        g.a_mb, g.a_mbs, g.a_sm, g.a_sk, g.a_shift, g.a_span = max(M, 1), 0, A.stride(0), A.stride(1), 0, INT_MAX
        g.k_inner, g.a_kbs, g.b_kbs = max(K, 1), 0, 0
        g.b_sk, g.b_sn = B.stride(0), B.stride(1)
        g.alpha = alpha
        split, kper = self._split_policy(M, N, K) if allow_split else (1, 0)
        return self._launch(g, M, N, K, out, bias, act, accumulate, split, kper)

    def linear(self, x, W, b=None, act=ACT_NONE, out=None, accumulate=False):
        """nn.Linear: x[M,K] @ W[N,K]^T + b."""
        return self.mm(x, W.t(), out=out, bias=b, act=act, accumulate=accumulate)

    def linear_logits(self, x, W, b, clf, need_direct=True):
        """The classifier head in ONE launch (icd_modules.py:159 + 177-182; the GEMM's fused row-dot epilogue, SURVEY 2.2
        K8/K9): direct = x @ W^T + b [M,D] and logits[m] = <relu(direct[m]), clf[m % L]> [M], rows m = b*L + l.
        need_direct=False (forward-only scoring) does not write `direct` at all.  Returns (direct | None, logits), or
        None where the epilogue has no form (D > 208 or D % 4: the caller runs linear + logits_fwd)."""
        _check_f32(x, W, b, clf)
        M, K = x.shape
        D = W.shape[0]
        L = clf.shape[0]
        if D > 208 or D % 4 or clf.shape[1] != D or not clf.is_contiguous() or M % L:
            return None
        if any(t.data_ptr() % 16 for t in (clf, b)):
            return None
        direct = torch.empty((M, D), dtype=torch.float32, device=x.device) if need_direct else None
        logits = torch.empty(M, dtype=torch.float32, device=x.device)
        Bt = W.t()
        g = GzslGemm()
        g.A, g.B = dptr(x), dptr(Bt)
        g.a_mb, g.a_mbs, g.a_sm, g.a_sk, g.a_shift, g.a_span = max(M, 1), 0, x.stride(0), x.stride(1), 0, INT_MAX
        g.k_inner, g.a_kbs, g.b_kbs = max(K, 1), 0, 0
        g.b_sk, g.b_sn = Bt.stride(0), Bt.stride(1)
        g.alpha = 1.0
        g.M, g.N, g.K = M, D, K
        # (without a store the C pointer is never dereferenced; it still has to be 16-byte aligned and non-null)
        g.C, g.ldc = dptr(direct if need_direct else logits), D
        g.bias, g.act, g.accumulate = ptr(b), ACT_NONE, 0
        g.split_k, g.k_per_split, g.slab_stride = 1, 0, 0
        g.dot_w, g.dot_out, g.dot_period, g.dot_no_store = dptr(clf), dptr(logits), L, 0 if need_direct else 1
        call('gzsl_gemm', g)
        return direct, logits

    def colsum(self, x2d, out=None):
        M, N = x2d.shape
        assert x2d.stride(1) == 1
        if out is None:
            out = torch.zeros(N, dtype=torch.float32, device=x2d.device)
        if M >= 4096 or self.deterministic:  # partial sums through a workspace instead of thousands of same-address atomics
            work = torch.empty(min(512, (M + 127) // 128) * N, dtype=torch.float32, device=x2d.device)
            call('gzsl_colsum_ws', dptr(x2d), ptr(out), M, N, x2d.stride(0), ptr(work), work.numel())
        else:
            call('gzsl_colsum', dptr(x2d), ptr(out), M, N, x2d.stride(0))
        return out

    def colsum_static_(self, x2d, out):
        """out[n] += sum_m x2d[m,n] through gzsl_colsum_static: no workspace tensor, nothing allocated (N <= 8192)."""
        M, N = x2d.shape
        assert x2d.stride(1) == 1 and out.numel() == N
        call('gzsl_colsum_static', dptr(x2d), ptr(out), M, N, x2d.stride(0))
        return out

    def act_bwd_(self, d, y, act):
        call('gzsl_act_bwd', ptr(d), ptr(y), d.numel(), act)
        return d

    # ------------------------------------------------------------------------------------ embedding + conv encoder
    def emb_gather(self, W, ids, mask, seed, p, row_offset=0):
        rows = ids.numel()
        E = W.shape[1]
        out = torch.empty(tuple(ids.shape) + (E,), dtype=torch.float32, device=W.device)
        th, sc = drop_params(p)
        call('gzsl_emb_gather_fwd', ptr(W), ptr(ids), ptr(mask), ptr(out), rows, E, row_offset, seed & 0xFFFFFFFF, th, sc)
        return out

    def emb_scatter_add_(self, dW, ids, mask, dOut, seed, p, pad_idx=0, row_offset=0):
        th, sc = drop_params(p)
        rows, E = ids.numel(), dW.shape[1]
        if self.deterministic:
            # fixed order: apply mask x dropout to the rows (emb_gather over the identity index computes exactly that),
            # then gather per WORD over the token rows sorted by (id, row) -- no atomics
            flat = ids.reshape(-1)
            g = self.emb_gather(dOut.reshape(rows, E), torch.arange(rows, device=flat.device), mask, seed, p, row_offset)
            keep = flat != pad_idx
            r = torch.arange(rows, device=flat.device)[keep]
            w = flat[keep]
            order = torch.argsort(w * rows + r, stable=True)
            w, r = w[order], r[order]
            rowptr = torch.zeros(dW.shape[0] + 1, dtype=torch.int64, device=flat.device)
            rowptr[1:] = torch.cumsum(torch.bincount(w, minlength=dW.shape[0]), 0)
            self.spmm((rowptr, r.contiguous(), torch.ones(r.numel(), dtype=torch.float32, device=flat.device)),
                      g.view(rows, E), out=dW, accumulate=True)
            return dW
        # many rows per id (word ids are Zipf-distributed): sum runs of equal ids inside 128-row chunks first
        name = 'gzsl_emb_scatter_add_grouped' if (rows >= 1024 and E <= 297) else 'gzsl_emb_scatter_add'
        call(name, ptr(dW), ptr(ids), ptr(mask), ptr(dOut), rows, E, pad_idx, row_offset, seed & 0xFFFFFFFF, th, sc)
        return dW

    def conv_relu(self, e, Wc, bc):
        """relu(conv1d(e^T, Wc) + bc)^T: e [B,T,E], Wc [F,E,k] -> X [B,T',F] (icd_modules.py:131-132)."""
        B, T, E = e.shape
        F_, _, k = Wc.shape
        Tp = T - k + 1
        Wp = torch.empty((F_, k * E), dtype=torch.float32, device=e.device)
        call('gzsl_conv_weight_pack', ptr(Wc), ptr(Wp), None, F_, E, k)
        X = torch.empty((B, Tp, F_), dtype=torch.float32, device=e.device)
        g = GzslGemm()
        g.A, g.B = dptr(e), dptr(Wp)
        g.a_mb, g.a_mbs, g.a_sm, g.a_sk, g.a_shift, g.a_span = Tp, T * E, E, 1, 0, INT_MAX
        g.k_inner, g.a_kbs, g.b_kbs = k * E, 0, 0
        g.b_sk, g.b_sn = 1, k * E
        g.alpha = 1.0
        self._launch(g, B * Tp, F_, k * E, X.view(B * Tp, F_), bc, ACT_RELU, False, 1, 0)
        return X

    def conv_bwd(self, dY, e, Wc, need_de=True):
        """dY [B,T',F] (already multiplied by the relu mask) -> (de [B,T,E] | None, dWc [F,E,k], dbc [F])."""
        B, T, E = e.shape
        F_, _, k = Wc.shape
        Tp = T - k + 1
        dev = e.device
        dY2 = dY.view(B * Tp, F_)
        dbc = self.colsum(dY2)
        # weight gradient, computed transposed ([k*E, F]) so that N = F uses the 208-wide tile
        dWc = torch.empty((F_, E, k), dtype=torch.float32, device=dev)
        g = GzslGemm()
        g.A, g.B = dptr(e), dptr(dY)
        M, N, K = k * E, F_, B * Tp
        g.a_mb, g.a_mbs, g.a_sm, g.a_sk, g.a_shift, g.a_span = M, 0, 1, E, 0, INT_MAX
        g.k_inner, g.a_kbs = Tp, T * E
        g.b_sk, g.b_sn, g.b_kbs = F_, 1, Tp * F_
        g.alpha = 1.0
        if Tp < 32:  # the in-kernel k decomposition needs k_inner >= 32: tiny notes go one note at a time
            dWc.zero_()
            for b in range(B):
                g2 = GzslGemm()
                g2.A, g2.B = dptr(e[b]), dptr(dY[b])
                g2.a_mb, g2.a_mbs, g2.a_sm, g2.a_sk, g2.a_shift, g2.a_span = M, 0, 1, E, 0, INT_MAX
                g2.k_inner, g2.a_kbs, g2.b_sk, g2.b_sn, g2.b_kbs, g2.alpha = Tp, 0, F_, 1, 0, 1.0
                self._launch(g2, M, N, Tp, dWc, None, ACT_NONE, True, 1, 0, unpack=(k, E))
        else:
            split, kper = self._split_policy(M, N, K)
            self._launch(g, M, N, K, dWc, None, ACT_NONE, False, split, kper, unpack=(k, E))
        de = None
        if need_de:
            Wflip = torch.empty((E, k * F_), dtype=torch.float32, device=dev)
            call('gzsl_conv_weight_pack', ptr(Wc), None, ptr(Wflip), F_, E, k)
            de = torch.empty((B, T, E), dtype=torch.float32, device=dev)
            g = GzslGemm()
            g.A, g.B = dptr(dY), dptr(Wflip)
            g.a_mb, g.a_mbs, g.a_sm, g.a_sk = T, Tp * F_, F_, 1
            g.a_shift, g.a_span = -(k - 1) * F_, Tp * F_
            g.k_inner, g.a_kbs, g.b_kbs = k * F_, 0, 0
            g.b_sk, g.b_sn = 1, k * F_
            g.alpha = 1.0
            self._launch(g, B * T, E, k * F_, de.view(B * T, E), None, ACT_NONE, False, 1, 0)
        return de, dWc, dbc

    # ------------------------------------------------------------------------------------ label side
    def label_avg(self, W, idx, mask):
        N, Td = idx.shape
        out = torch.empty((N, W.shape[1]), dtype=torch.float32, device=W.device)
        call('gzsl_label_avg_fwd', ptr(W), ptr(idx), ptr(mask), ptr(out), N, Td, W.shape[1])
        return out

    def label_avg_bwd_(self, dW, davg, idx, mask, pad_idx=0, inv=None):
        """dW[idx[n,j]] += mask[n,j] / sum(mask[n]) * davg[n].  inv = GraphBuffers.desc_inverse(V): the same sum as a
        gather over the inverted index, one float atomic per (word, column)."""
        N, Td = idx.shape
        if inv is not None:
            rowptr, col, val = inv
            call('gzsl_spmm_csr', ptr(rowptr), ptr(col), ptr(val), ptr(davg), None, None, ptr(dW), dW.shape[0],
                 dW.shape[1], 1 if self.deterministic else 2)
            return dW
        call('gzsl_label_avg_bwd', ptr(dW), ptr(davg), ptr(idx), ptr(mask), N, Td, dW.shape[1], pad_idx)
        return dW

    def spmm(self, csr, x, pre=None, post=None, out=None, accumulate=False):
        rowptr, col, val = csr
        N, H = rowptr.numel() - 1, x.shape[1]   # output rows = CSR rows (x may have a different row count)
        if out is None:
            assert not accumulate
            out = torch.empty((N, H), dtype=torch.float32, device=x.device)
        call('gzsl_spmm_csr', ptr(rowptr), ptr(col), ptr(val), ptr(x), ptr(pre), ptr(post), ptr(out), N, H,
             int(accumulate))
        return out

    def gru_gate_fwd(self, gi, gh, h):
        hn = torch.empty_like(h)
        call('gzsl_gru_gate_fwd', ptr(gi), ptr(gh), ptr(h), ptr(hn), h.shape[0], h.shape[1])
        return hn

    def gru_gate_bwd(self, dhn, gi, gh, h):
        dgi, dgh, dh = torch.empty_like(gi), torch.empty_like(gh), torch.empty_like(h)
        call('gzsl_gru_gate_bwd', ptr(dhn), ptr(gi), ptr(gh), ptr(h), ptr(dgi), ptr(dgh), ptr(dh), h.shape[0],
             h.shape[1])
        return dgi, dgh, dh

    def gru_seq_step_(self, gi_all, t, gh, h, mask, pool, pool_mode):
        """Step t of the packed label GRU (inference): gi_all [Bg,T,3H], gh [Bg,3H]; updates h (and pool) in place."""
        Bg, T, H3 = gi_all.shape
        call('gzsl_gru_seq_step', dptr(gi_all[:, t]), T * H3, ptr(gh), ptr(h), ptr(mask), T, t, ptr(pool),
             {None: 0, 'last': 0, 'max': 1, 'mean': 2}[pool_mode], Bg, H3 // 3)

    # ------------------------------------------------------------------------------------ attention + head + loss
    def attn_fwd(self, PX, X, Q, l_offset, L_total, seed, p, out=None):
        """out = (O [B,L,F], lse [B,L]) contiguous views to fill (a note block of a larger batch), else allocated."""
        B, Tp, E = PX.shape
        F_ = X.shape[2]
        L = Q.shape[0]
        if out is None:
            O = torch.empty((B, L, F_), dtype=torch.float32, device=PX.device)
            lse = torch.empty((B, L), dtype=torch.float32, device=PX.device)
        else:
            O, lse = out
            assert tuple(O.shape) == (B, L, F_) and tuple(lse.shape) == (B, L)
        th, sc = drop_params(p)
        if self.attn_bf16x6 and E <= 256 and F_ <= 208:
            # EXPERIMENTAL (GZSL_ATTN_BF16X6=1): fp32 products from six bf16 MFMAs each (csrc/attn_bf16.hip)
            n = hip.load().gzsl_label_attn_fwd_bf16x6_work_bytes(B, Tp, E, F_)
            work = torch.empty(n, dtype=torch.uint8, device=PX.device)
            call('gzsl_label_attn_fwd_bf16x6', ptr(PX), ptr(X), ptr(Q), ptr(O), ptr(lse), B, Tp, E, F_, L, l_offset,
                 L_total, seed & 0xFFFFFFFF, th, sc, ptr(work), n)
            return O, lse
        call('gzsl_label_attn_fwd', ptr(PX), ptr(X), ptr(Q), ptr(O), ptr(lse), B, Tp, E, F_, L, l_offset, L_total,
             seed & 0xFFFFFFFF, th, sc)
        return O, lse

    def attn_bwd(self, dO, O, PX, X, Q, lse, l_offset, L_total, seed, p, dPX, dX, dQ, nsplit=None, delta=None):
        """Accumulates into dPX [B,Tp,E], dX [B,Tp,F], dQ [L,E].  delta [B,L] = rowsum(dO * O) when the caller already
        has it (logits_bwd_delta); computed here from dO and O otherwise."""
        B, Tp, E = PX.shape
        F_ = X.shape[2]
        L = Q.shape[0]
        if delta is None:
            delta = torch.empty((B, L), dtype=torch.float32, device=PX.device)
            call('gzsl_attn_delta', ptr(dO), ptr(O), ptr(delta), B * L, F_)
        assert tuple(delta.shape) == (B, L)
        if nsplit is None:
            tok_per_wg = 128 if E <= 100 else 64
            wgs = B * ((Tp + tok_per_wg - 1) // tok_per_wg)
            nsplit = 1
            while wgs * nsplit < 1024 and nsplit * 2 * 16 <= L:
                nsplit *= 2
        self.attn_bwd_main(dO, PX, X, Q, lse, delta, dPX, dX, dQ, l_offset, L_total, nsplit, seed, p)

    def attn_bwd_main(self, dO, PX, X, Q, lse, delta, dPX, dX, dQ, l_offset, L_total, nsplit, seed, p):
        """The label_attn_bwd kernel alone (bench.py times this launch for the roofline line)."""
        B, Tp, E = PX.shape
        th, sc = drop_params(p)
        if self.deterministic:
            n = hip.load().gzsl_label_attn_bwd_det_work_floats(B, Tp, E, X.shape[2], Q.shape[0], nsplit)
            free = torch.cuda.mem_get_info(PX.device)[0] + torch.cuda.memory_reserved(PX.device) - torch.cuda.memory_allocated(PX.device)
            if 4 * n > free:
                raise RuntimeError(f'deterministic attention backward needs a {4 * n / 2**30:.2f} GiB workspace '
                                   f'(B*ceil(Tp/{128 if E <= 100 else 64})*L*E + nsplit*B*Tp*(E+F) floats; '
                                   f'{free / 2**30:.2f} GiB free): lower the batch or the label shard')
            work = torch.empty(n, dtype=torch.float32, device=PX.device)
            call('gzsl_label_attn_bwd_det', ptr(dO), ptr(PX), ptr(X), ptr(Q), ptr(lse), ptr(delta), ptr(dPX), ptr(dX),
                 ptr(dQ), B, Tp, E, X.shape[2], Q.shape[0], l_offset, L_total, nsplit, seed & 0xFFFFFFFF, th, sc,
                 ptr(work), n)
            return
        if self.attn_bf16x6 and E <= 112 and X.shape[2] <= 208:
            # opt-in (GZSL_ATTN_BF16X6=1): fp32 products from six bf16 MFMAs each (csrc/attn_bwd_bf16.hip); 64-token workgroups
            F_, L = X.shape[2], Q.shape[0]
            n = hip.load().gzsl_label_attn_bwd_bf16x6_work_bytes(B, Tp, E, F_, L)
            work = torch.empty(n, dtype=torch.uint8, device=PX.device)
            wgs = B * ((Tp + 63) // 64)
            ns = 1
            while wgs * ns < 1024 and ns * 2 * 32 <= L:
                ns *= 2
            ns = int(os.environ.get('GZSL_BB_NSPLIT', ns))  # (diagnostic override)
            defer = bool(getattr(self, 'defer_dq', False))
            call('gzsl_label_attn_bwd_bf16x6', ptr(dO), ptr(PX), ptr(X), ptr(Q), ptr(lse), ptr(delta), ptr(dPX), ptr(dX),
                 ptr(dQ), B, Tp, E, F_, L, l_offset, L_total, ns, seed & 0xFFFFFFFF, th, sc, ptr(work), n, int(defer))
            if defer:  # the engine adds the dQ slabs later, on the stream that runs the label-side backward
                self._pending_dq.append((work, dQ, (B, Tp, E, F_, L)))
            return
        call('gzsl_label_attn_bwd', ptr(dO), ptr(PX), ptr(X), ptr(Q), ptr(lse), ptr(delta), ptr(dPX), ptr(dX),
             ptr(dQ), B, Tp, E, X.shape[2], Q.shape[0], l_offset, L_total, nsplit, seed & 0xFFFFFFFF, th, sc)

    def flush_pending_dq(self):
        """Adds the dQ slabs of deferred bf16x6 attention backwards (defer_dq) on the CURRENT stream."""
        pend, self._pending_dq = self._pending_dq, []
        for work, dQ, dims in pend:
            call('gzsl_label_attn_bwd_bf16x6_dq_reduce', ptr(work), ptr(dQ), *dims)
        return pend  # the caller keeps the workspaces alive until its streams have joined

    def logits_fwd(self, direct, clf):
        B, L, D = direct.shape
        logits = torch.empty((B, L), dtype=torch.float32, device=direct.device)
        call('gzsl_logits_fwd', ptr(direct), ptr(clf), ptr(logits), B, L, D)
        return logits

    def logits_bwd(self, direct, clf, dlogits, need_ddirect=True):
        B, L, D = direct.shape
        ddirect = torch.empty_like(direct) if need_ddirect else None
        dclf = torch.empty((L, D), dtype=torch.float32, device=direct.device)
        call('gzsl_logits_bwd', ptr(direct), ptr(clf), ptr(dlogits), ptr(ddirect), ptr(dclf), B, L, D, 0)
        return ddirect, dclf

    def logits_bwd_delta(self, direct, clf, bo, dlogits):
        """(ddirect, dclf, delta): logits_bwd plus the attention backward's row constant delta [B,L] from `direct`
        and the output_proj bias alone (gzsl_logits_bwd_delta).  None when the shape has no vector path (D % 4)."""
        B, L, D = direct.shape
        if D % 4 or D > 1024:
            return None
        ddirect = torch.empty_like(direct)
        dclf = torch.empty((L, D), dtype=torch.float32, device=direct.device)
        delta = torch.empty((B, L), dtype=torch.float32, device=direct.device)
        call('gzsl_logits_bwd_delta', ptr(direct), ptr(clf), ptr(bo), ptr(dlogits), ptr(ddirect), ptr(dclf), ptr(delta),
             B, L, D)
        return ddirect, dclf, delta

    def ldam_bce(self, logits, y, margin, C, inv_b, need_grad=True, need_probs=False):
        B, L = logits.shape
        # the kernel reads y with the logits' row pitch and margin[l] for l < L: a wider y (or label_indices beyond
        # eval_code_size) would silently pair targets with the wrong columns where the reference raises a shape error
        if tuple(y.shape) != (B, L):
            raise RuntimeError(f'ldam_bce: targets {tuple(y.shape)} do not match logits {(B, L)}')
        if margin is not None and margin.numel() != L:
            raise RuntimeError(f'ldam_bce: {margin.numel()} margins for {L} labels')
        loss = torch.zeros((), dtype=torch.float32, device=logits.device)
        dlogits = torch.empty_like(logits) if need_grad else None
        probs = torch.empty_like(logits) if need_probs else None
        use_margin = margin is not None and C > 0
        call('gzsl_ldam_bce', ptr(logits), ptr(y), ptr(margin) if use_margin else None, float(C), float(inv_b),
             ptr(loss), ptr(dlogits), ptr(probs), B, L)
        return loss, dlogits, probs

    # ------------------------------------------------------------------------------------ batching / eval scoring
    def multihot(self, y_indices, L):
        """[B,W] int64 label lists padded with -1 -> dense fp32 [B,L] targets (utils/helper.py:117-123)."""
        B, W = y_indices.shape
        y = torch.empty((B, L), dtype=torch.float32, device=y_indices.device)
        call('gzsl_multihot', ptr(y_indices), W, ptr(y), B, L)
        return y

    def eval_counts_(self, logits, y, counts, probs=None):
        """counts [4,Lc] int32 += per-label (yhat&y, yhat, y, yhat|y) over the rows of logits[:, :Lc] / y[:, :Lc]."""
        B = logits.shape[0]
        Lc = counts.shape[1]
        assert counts.dtype == torch.int32 and counts.shape[0] == 4
        assert logits.stride(1) == 1 and y.stride(1) == 1 and logits.shape[1] >= Lc and y.shape[1] >= Lc
        assert logits.dtype == torch.float32 and y.dtype == torch.float32
        if probs is not None:
            assert probs.shape == (B, Lc) and probs.stride(1) == 1 and probs.dtype == torch.float32
        if not logits.is_cuda:
            raise RuntimeError('gzsl_text_amd kernels need tensors on a HIP device; there is no CPU path')
        call('gzsl_eval_counts', dptr(logits), logits.stride(0), dptr(y), y.stride(0), ptr(counts),
             None if probs is None else dptr(probs), 0 if probs is None else probs.stride(0), B, Lc)
        return counts

    def forward_final(self, fc, feats, labels, margin, C, scale, need_grad=True):
        Bs, D = feats.shape
        loss = torch.zeros((), dtype=torch.float32, device=feats.device)
        dfc = torch.zeros_like(fc) if need_grad else None
        use_margin = margin is not None and C > 0
        call('gzsl_forward_final', ptr(fc), ptr(feats), ptr(labels), ptr(margin) if use_margin else None, float(C),
             float(scale), ptr(loss), ptr(dfc), None, Bs, D)
        return loss, dfc

    def adamw_step_(self, p, g, m, v, lr, beta1, beta2, eps, wd):
        call('gzsl_adamw_nobias_step', ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), lr, beta1, beta2, eps, wd)

    # ------------------------------------------------------------------------------------ WGAN-GP pieces
    def gp_interp(self, real, fake, alpha):
        out = torch.empty_like(real)
        call('gzsl_gp_interp', ptr(real), ptr(fake), ptr(alpha), ptr(out), real.shape[0], real.shape[1])
        return out

    def gp_slope_v(self, h, w2):
        v, s = torch.empty_like(h), torch.empty_like(h)
        call('gzsl_gp_slope_v', ptr(h), ptr(w2), ptr(v), ptr(s), h.shape[0], h.shape[1])
        return v, s

    def gp_rownorm_(self, g, lam, out=None):
        """g <- d gp / d g in place; the penalty is ADDED to `out` (a zeroed 1-element tensor; allocated when None)."""
        gp = torch.zeros((), dtype=torch.float32, device=g.device) if out is None else out
        call('gzsl_gp_rownorm', ptr(g), ptr(gp), g.shape[0], g.shape[1], float(lam))
        return gp

    def gp_dw2_(self, dv, s, dw2):
        call('gzsl_gp_dw2', ptr(dv), ptr(s), ptr(dw2), dv.shape[0], dv.shape[1])

    def rowdot(self, a, w, bias, lrelu_in=False):
        out = torch.empty(a.shape[0], dtype=torch.float32, device=a.device)
        call('gzsl_rowdot', ptr(a), ptr(w), ptr(bias), ptr(out), a.shape[0], a.shape[1], int(lrelu_in))
        return out

    def weighted_sum(self, x, w, scale):
        out = torch.zeros((), dtype=torch.float32, device=x.device)
        call('gzsl_weighted_sum', ptr(x), ptr(w), float(scale), ptr(out), x.numel())
        return out

    def rowdot_bwd(self, dout, w, act, dw, db, need_da=True, lrelu_in=False):
        da = torch.empty_like(act) if need_da else None
        call('gzsl_rowdot_bwd', ptr(dout), ptr(w), ptr(act), ptr(da), ptr(dw), ptr(db), act.shape[0], act.shape[1],
             int(lrelu_in))
        return da

    def gan_critic_head(self, hf, lh, w2, sim, alpha, dw2, db1, cost):
        """gzsl_gan_critic_head: returns (lst [3Bg,H], dhs [Bg,H], sl [Bg,H]); dw2 / db1 / cost are accumulated."""
        Bg, H = lh.shape
        assert hf.shape == (2 * Bg, H) and alpha.numel() == Bg and (sim is None or sim.numel() == Bg)
        lst = torch.empty((3 * Bg, H), dtype=torch.float32, device=lh.device)
        dhs, sl = torch.empty_like(lh), torch.empty_like(lh)
        call('gzsl_gan_critic_head', ptr(hf), ptr(lh), ptr(w2), ptr(sim), ptr(alpha), ptr(lst), ptr(dhs), ptr(sl), ptr(dw2),
             ptr(db1), ptr(cost), Bg, H)
        return lst, dhs, sl

    def gan_gen_head(self, hf, lh, w2, b2, sim, cost):
        Bg, H = lh.shape
        assert hf.shape == (Bg, H) and (sim is None or sim.numel() == Bg)
        dh = torch.empty_like(lh)
        call('gzsl_gan_gen_head', ptr(hf), ptr(lh), ptr(w2), ptr(b2), ptr(sim), ptr(dh), ptr(cost), Bg, H)
        return dh

    ADAM_STATE_INTS = 1024  # GZSL_ADAM_STATE_INTS

    def adam_flat_step_(self, p, g, m, v, state, lr, beta1, beta2, eps):
        """torch.optim.Adam.step over one flat buffer; `state` = int32 [ADAM_STATE_INTS] on the device, zero before the
        first step (per-workgroup copies of the step count; state[0] = steps taken)."""
        assert p.numel() == g.numel() == m.numel() == v.numel()
        assert state.dtype == torch.int32 and state.numel() == self.ADAM_STATE_INTS
        call('gzsl_adam_flat_step', ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), lr, beta1, beta2, eps, ptr(state))

    def keyword_ce_sim_(self, logits, kw_idx, kw_mask, sim, grad_scale, loss_scale, loss_sum):
        """keyword_ce_ with train_gan.py:288-296's row weights computed from `sim` inside the kernel; the weighted loss
        sum is ADDED to `loss_sum` (1 element)."""
        Bg, Vk = logits.shape
        loss_b = torch.empty(Bg, dtype=torch.float32, device=logits.device)
        call('gzsl_keyword_ce_sim', ptr(logits), ptr(kw_idx), ptr(kw_mask), ptr(sim), float(grad_scale), float(loss_scale),
             ptr(loss_b), ptr(loss_sum), Bg, Vk, kw_idx.shape[1])
        return loss_b

    def keyword_ce_(self, logits, kw_idx, kw_mask, w):
        """Returns per-row loss; overwrites `logits` with w_b * d loss_b / d logits."""
        Bg, Vk = logits.shape
        loss_b = torch.empty(Bg, dtype=torch.float32, device=logits.device)
        call('gzsl_keyword_ce', ptr(logits), ptr(kw_idx), ptr(kw_mask), ptr(w), ptr(loss_b), Bg, Vk, kw_idx.shape[1])
        return loss_b
