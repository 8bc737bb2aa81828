"""Forward / backward orchestration of the base feature extractor on top of an `ops` provider.

This is the hand-written replacement for what autograd did in the reference
(`loss.mean().backward()`, train_base.py:112): every function below launches HIP kernels through
`ops` (gzsl_text_amd.ops_hip.HipOps) and keeps exactly the tensors backward needs.  The nn.Module
mirrors in gzsl_text_amd/modules call it from two torch.autograd.Functions.

Parameter names are the reference's state-dict keys (icd_modules.py:36-90):
  emb.embed.weight [V,E], conv_modules.0.{weight [F,E,k], bias}, proj.{weight [E,F], bias},
  output_proj.{weight [D,F], bias}, graph_label_encoder.gru.{weight_ih, weight_hh, bias_ih, bias_hh}
  (or graph_label_encoder.gc{1,2}.fc.{weight,bias} for --graph_encoder=conv).
"""
import torch

from .hip import ACT_NONE, ACT_RELU, ACT_TANH

P_EMB_DROP = 0.5   # icd_modules.py:55
P_ATTN_DROP = 0.2  # icd_modules.py:59
EPS_DEG = 1e-7     # graph_module.py:117
_NO_FUSED_HEAD = __import__('os').environ.get('GZSL_NO_FUSED_HEAD') == '1'  # diagnostic: output_proj and logits as two launches


def build_csr(adj):
    """torch dense / sparse-COO [N,N] adjacency -> (csr, csr_T); each csr = (rowptr int64 [N+1], col int64, val fp32).

    One-off setup work (the reference re-does a sparse matmul from the COO tensor every step,
    graph_module.py:114); it runs wherever `adj` lives.
    """
    if adj.is_sparse:
        a = adj.coalesce()
        idx, val = a.indices(), a.values()
    else:
        idx = adj.nonzero().t().contiguous()
        val = adj[idx[0], idx[1]]
    N = adj.shape[0]
    out = []
    for r, c in ((idx[0], idx[1]), (idx[1], idx[0])):
        order = torch.argsort(r * N + c)
        rs, cs, vs = r[order], c[order], val[order]
        rowptr = torch.zeros(N + 1, dtype=torch.int64, device=adj.device)
        rowptr[1:] = torch.cumsum(torch.bincount(rs, minlength=N), 0)
        out.append((rowptr.contiguous(), cs.contiguous().to(torch.int64), vs.contiguous().to(torch.float32)))
    return out[0], out[1]


class GraphBuffers:
    """Device-resident, kernel-ready form of the four registered buffers (icd_modules.py:46-49)."""

    def __init__(self, code_idx_matrix, code_idx_mask, adj_matrix, num_neighbors):
        self.idx = code_idx_matrix.contiguous()
        self.cmask = code_idx_mask.contiguous().to(torch.float32)
        self.csr, self.csr_t = build_csr(adj_matrix)
        self.inv_deg = (1.0 / (num_neighbors.to(torch.float32) + EPS_DEG)).contiguous()
        self.N = self.idx.shape[0]
        self._desc_inv = {}

    def desc_inverse(self, V, pad_idx=0):
        """Inverted description index as a CSR over the V embedding rows: row w lists the labels n whose description
        holds word w, with value mask[n,j] / sum_j mask[n,j] (one entry per occurrence).  The adjoint of
        average_label_embedding (icd_modules.py:110-113) is then a gather per WORD -- dW[w] += sum val * davg[n] -- instead
        of 36 M element atomics at cfg2.  One-off setup work (the buffers are registered constants)."""
        key = (int(V), int(pad_idx))
        if key not in self._desc_inv:
            N, Td = self.idx.shape
            coef = self.cmask / self.cmask.sum(1, keepdim=True)
            w = self.idx.reshape(-1)
            n = torch.arange(N, device=w.device).repeat_interleave(Td)
            keep = (w != pad_idx) & (self.cmask.reshape(-1) != 0)
            w, n, v = w[keep], n[keep], coef.reshape(-1)[keep]
            order = torch.argsort(w * N + n, stable=True)
            w, n, v = w[order], n[order], v[order]
            rowptr = torch.zeros(V + 1, dtype=torch.int64, device=w.device)
            rowptr[1:] = torch.cumsum(torch.bincount(w, minlength=V), 0)
            self._desc_inv[key] = (rowptr.contiguous(), n.contiguous().to(torch.int64), v.contiguous().to(torch.float32))
        return self._desc_inv[key]


class BaseEngine:
    def __init__(self, ops, graph_encoder='gate', steps=2):
        self.ops = ops
        self.graph_encoder = graph_encoder
        self.steps = steps

    def _deferring(self):
        """While active, the bf16x6 attention backward leaves its dQ slab sum to `label_bwd` (every model backward ends
        with one: _BaseModelFn.backward).  No effect on providers without the switch (exact-fp32 kernels, CPU stand-in)."""
        import contextlib
        ops = self.ops

        @contextlib.contextmanager
        def cm():
            had = getattr(ops, 'defer_dq', None)
            if had is not None:
                ops.defer_dq = True
            try:
                yield
            finally:
                if had is not None:
                    ops.defer_dq = had
        return cm()

    # ------------------------------------------------------------------------------------------ label side
    def label_fwd(self, P, gb, keep=True):
        """icd_modules.py:115-123 -> avg [N,E], graph [N,D], cache."""
        ops = self.ops
        avg = ops.label_avg(P['emb.embed.weight'], gb.idx, gb.cmask)
        cache = None
        if self.graph_encoder == 'gate':
            pre = 'graph_label_encoder.gru.'
            h = avg
            steps = []
            for _ in range(self.steps):
                u = ops.spmm(gb.csr, h, post=gb.inv_deg)
                gi = ops.linear(u, P[pre + 'weight_ih'], P[pre + 'bias_ih'])
                gh = ops.linear(h, P[pre + 'weight_hh'], P[pre + 'bias_hh'])
                hn = ops.gru_gate_fwd(gi, gh, h)
                if keep:
                    steps.append((u, h, gi, gh))
                h = hn
            graph = torch.cat([avg, h], 1)
            cache = steps
        elif self.graph_encoder == 'conv':
            f = avg
            steps = []
            for name in ('gc1', 'gc2'):
                nb = ops.spmm(gb.csr, f, post=gb.inv_deg)
                cat = torch.cat([f, nb], 1)
                out = ops.linear(cat, P[f'graph_label_encoder.{name}.fc.weight'],
                                 P[f'graph_label_encoder.{name}.fc.bias'], act=ACT_RELU)
                if keep:
                    steps.append((cat, out, f.shape[1]))
                f = out
            graph = torch.cat([f, avg], 1)
            cache = steps
        else:
            graph = avg
        return avg, graph, cache

    def label_bwd(self, davg, dgraph, cache, P, gb, G):
        """Adds the label-side parameter gradients into G; davg / dgraph are [N,E] / [N,D] (rows past L zero)."""
        ops = self.ops
        E = davg.shape[1]
        # deferred dQ sums of the bf16x6 attention backward (attn_bwd_block / head_bwd): added here, on the stream that
        # runs the label side, beside the note encoder's backward (which does not read davg)
        self._keep = getattr(ops, 'flush_pending_dq', lambda: None)()
        if self.graph_encoder == 'gate':
            pre = 'graph_label_encoder.gru.'
            dx = dgraph[:, :E].contiguous()
            dh = dgraph[:, E:].contiguous()
            for u, h, gi, gh in reversed(cache):
                dgi, dgh, dh_direct = ops.gru_gate_bwd(dh, gi, gh, h)
                ops.mm(dgi.t(), u, out=G[pre + 'weight_ih'], accumulate=True)
                ops.mm(dgh.t(), h, out=G[pre + 'weight_hh'], accumulate=True)
                ops.colsum(dgi, out=G[pre + 'bias_ih'])
                ops.colsum(dgh, out=G[pre + 'bias_hh'])
                du = ops.mm(dgi, P[pre + 'weight_ih'])
                dh = ops.mm(dgh, P[pre + 'weight_hh'], out=dh_direct, accumulate=True)
                ops.spmm(gb.csr_t, du, pre=gb.inv_deg, out=dh, accumulate=True)
            davg = davg + dx + dh
        elif self.graph_encoder == 'conv':
            H = dgraph.shape[1] - E
            df = dgraph[:, :H].contiguous()
            dx = dgraph[:, H:].contiguous()
            for name, (cat, out, din) in zip(('gc2', 'gc1'), reversed(cache)):
                W = P[f'graph_label_encoder.{name}.fc.weight']
                ops.act_bwd_(df, out, ACT_RELU)
                ops.mm(df.t(), cat, out=G[f'graph_label_encoder.{name}.fc.weight'], accumulate=True)
                ops.colsum(df, out=G[f'graph_label_encoder.{name}.fc.bias'])
                dcat = ops.mm(df, W)
                dnb = dcat[:, din:].contiguous()
                df = dcat[:, :din].contiguous()
                ops.spmm(gb.csr_t, dnb, pre=gb.inv_deg, out=df, accumulate=True)
            davg = davg + dx + df
        else:
            davg = davg + dgraph
        ops.label_avg_bwd_(G['emb.embed.weight'], davg.contiguous(), gb.idx, gb.cmask,
                           inv=gb.desc_inverse(G['emb.embed.weight'].shape[0]))

    # ------------------------------------------------------------------------------------------ note encoder
    def encoder_fwd(self, P, x, mask, seed, training, row_offset=0):
        """icd_modules.py:153-157 + 130-135 -> X [B,T',F] (values), PX [B,T',E] (keys), e [B,T,E]."""
        ops = self.ops
        e = ops.emb_gather(P['emb.embed.weight'], x, mask, seed, P_EMB_DROP if training else 0.0, row_offset)
        X = ops.conv_relu(e, P['conv_modules.0.weight'], P['conv_modules.0.bias'])
        B, Tp, F_ = X.shape
        PX = ops.linear(X.view(B * Tp, F_), P['proj.weight'], P['proj.bias'], act=ACT_TANH).view(B, Tp, -1)
        return X, PX, e

    def encoder_bwd(self, dX, dPX, X, PX, e, x, mask, seed, training, P, G, row_offset=0):
        """dX / dPX are consumed (modified in place).  Adds conv / proj / embedding gradients into G."""
        ops = self.ops
        B, Tp, F_ = X.shape
        E = PX.shape[2]
        ops.act_bwd_(dPX, PX, ACT_TANH)
        dpre = dPX.view(B * Tp, E)
        X2 = X.view(B * Tp, F_)
        ops.mm(dpre.t(), X2, out=G['proj.weight'], accumulate=True)
        ops.colsum(dpre, out=G['proj.bias'])
        ops.mm(dpre, P['proj.weight'], out=dX.view(B * Tp, F_), accumulate=True)
        ops.act_bwd_(dX, X, ACT_RELU)
        de, dWc, dbc = ops.conv_bwd(dX, e, P['conv_modules.0.weight'])
        G['conv_modules.0.weight'].add_(dWc)
        G['conv_modules.0.bias'].add_(dbc)
        ops.emb_scatter_add_(G['emb.embed.weight'], x, mask, de, seed, P_EMB_DROP if training else 0.0,
                             row_offset=row_offset)

    # ------------------------------------------------------------------------------------------ attention + head
    def attn_fwd_block(self, PX, X, Q, l_offset, L_total, seed, training, O, lse, c0):
        """Attention of the note block [c0, c0 + len(PX)) of a batch into O[c0:], lse[c0:].  The dropout hash is keyed
        on row = note * L_total + label, so a block's global note index travels in the label offset."""
        n = PX.shape[0]
        self.ops.attn_fwd(PX, X, Q, l_offset + c0 * L_total, L_total, seed, P_ATTN_DROP if training else 0.0,
                          out=(O[c0:c0 + n], lse[c0:c0 + n]))

    def attn_bwd_block(self, dO, O, lse, PX, X, Q, c0, c1, l_offset, L_total, seed, training, dPX, dX, dQ, delta=None):
        """Backward of `attn_fwd_block` for notes [c0, c1): adds into dPX / dX (tensors of THAT block) and dQ.
        delta [B,L]: rowsum(dO * O) of the whole batch when head_bwd_pre has produced it."""
        with self._deferring():
            self.ops.attn_bwd(dO[c0:c1], O[c0:c1], PX, X, Q, lse[c0:c1], l_offset + c0 * L_total, L_total, seed,
                              P_ATTN_DROP if training else 0.0, dPX, dX, dQ, delta=None if delta is None else delta[c0:c1])

    def head_fwd(self, P, X, PX, Q, clf, l_offset, L_total, seed, training, attn=None, need_direct=True):
        """icd_modules.py:137-142,159,177-182 for the label rows [l_offset, l_offset + len(Q)).  attn = (O, lse) when
        the caller has already run the attention (note block by note block)."""
        ops = self.ops
        if attn is None:
            O, lse = ops.attn_fwd(PX, X, Q, l_offset, L_total, seed, P_ATTN_DROP if training else 0.0)
        else:
            O, lse = attn
        B, L, F_ = O.shape
        # output_proj + relu + per-label dot in one launch where the GEMM's fused epilogue has a form for it; with
        # need_direct=False (forward-only scoring) `direct` [B,L,D] is never written
        fused = None if _NO_FUSED_HEAD else getattr(ops, 'linear_logits', None)
        out = fused(O.view(B * L, F_), P['output_proj.weight'], P['output_proj.bias'], clf, need_direct) if fused else None
        if out is not None:
            direct, logits = out
            return O, lse, (None if direct is None else direct.view(B, L, -1)), logits.view(B, L)
        direct = ops.linear(O.view(B * L, F_), P['output_proj.weight'], P['output_proj.bias']).view(B, L, -1)
        logits = ops.logits_fwd(direct, clf)
        return O, lse, direct, logits

    def head_bwd_pre(self, dlogits, O, direct, clf, P):
        """Classifier + output_proj data gradient: returns (dclf [L,D], dO [B,L,F], pending weight-gradient operands,
        delta [B,L] = rowsum(dO * O) for the attention backward, or None when it has to be computed from dO and O).
        delta comes out of the pass over `direct` that produces ddirect: dO = ddirect.Wo and direct = O.Wo^T + bo, so
        rowsum(dO * O) = rowsum(ddirect * (direct - bo)) -- no separate pass over dO and O (2 x 115 MB at cfg2)."""
        ops = self.ops
        B, L, F_ = O.shape
        D = direct.shape[2]
        fused = ops.logits_bwd_delta(direct, clf, P['output_proj.bias'], dlogits)
        if fused is None:
            ddirect, dclf = ops.logits_bwd(direct, clf, dlogits, need_ddirect=True)
            delta = None
        else:
            ddirect, dclf, delta = fused
        dd2 = ddirect.view(B * L, D)
        dO = ops.mm(dd2, P['output_proj.weight']).view(B, L, F_)
        return dclf, dO, (dd2, O.view(B * L, F_)), delta

    def head_bwd(self, dlogits, O, lse, direct, X, PX, Q, clf, l_offset, L_total, seed, training, P, G,
                 dX, dPX, dQ, need_encoder_grads=True, defer_wgrad=False):
        """Returns dclf [L,D]; accumulates output_proj grads into G and (optionally) dX, dPX, dQ.

        defer_wgrad=True returns (dclf, pending) instead and leaves the output_proj weight / bias gradients to
        `head_wgrad(pending, G)`: they depend on nothing downstream, so the sharded path runs them while the
        reduce-scatter of dX / dPX is on the wire."""
        ops = self.ops
        if not need_encoder_grads:
            _, dclf = ops.logits_bwd(direct, clf, dlogits, need_ddirect=False)
            return (dclf, None) if defer_wgrad else dclf
        dclf, dO, pending, delta = self.head_bwd_pre(dlogits, O, direct, clf, P)
        if not defer_wgrad:
            self.head_wgrad(pending, G)
        with self._deferring():
            ops.attn_bwd(dO, O, PX, X, Q, lse, l_offset, L_total, seed, P_ATTN_DROP if training else 0.0, dPX, dX, dQ,
                         delta=delta)
        return (dclf, pending) if defer_wgrad else dclf

    def head_wgrad(self, pending, G):
        if pending is None:
            return
        dd2, O2 = pending
        self.ops.mm(dd2.t(), O2, out=G['output_proj.weight'], accumulate=True)
        self.ops.colsum(dd2, out=G['output_proj.bias'])


def zeros_like_params(P, names):
    return {k: torch.zeros_like(P[k]) for k in names}
