"""CPU stand-in for gzsl_text_amd.ops_hip.HipOps -- TEST INFRASTRUCTURE, never imported by the product.

Implements the same tensor-level interface with plain torch fp32 ops (and the oracle's counter RNG), so the
engine's orchestration -- autograd wiring, what is saved for backward, label sharding and its collectives -- can be
exercised on CPU (`-m "not gpu"`, gloo world_size 2).  Each method is the "plain PyTorch fp32 reference of the same
op" the GPU parity tests compare the HIP kernels with.
"""
import numpy as np
import torch
import torch.nn.functional as F

from oracle import rng

ACT_NONE, ACT_RELU, ACT_TANH, ACT_LRELU = 0, 1, 2, 3


def _act(x, act):
    if act == ACT_RELU:
        return torch.relu(x)
    if act == ACT_TANH:
        return torch.tanh(x)
    if act == ACT_LRELU:
        return F.leaky_relu(x, 0.2)
    return x


class CpuOps:
    name = 'cpu-test-standin'

    def mm(self, A, B, out=None, bias=None, act=ACT_NONE, accumulate=False, alpha=1.0, allow_split=True):
        r = alpha * (A @ B)
        if bias is not None:
            r = r + bias
        if accumulate:
            r = r + out
        r = _act(r, act)
        if out is None:
            return r.contiguous()
        out.copy_(r)
        return out

    def linear(self, x, W, b=None, act=ACT_NONE, out=None, accumulate=False):
        return self.mm(x, W.t(), out=out, bias=b, act=act, accumulate=accumulate)

    def colsum(self, x2d, out=None):
        s = x2d.sum(0)
        if out is None:
            return s
        out += s
        return out

    def act_bwd_(self, d, y, act):
        if act == ACT_RELU:
            d.mul_((y > 0).float())
        elif act == ACT_TANH:
            d.mul_(1 - y * y)
        elif act == ACT_LRELU:
            d.mul_(torch.where(y > 0, torch.ones_like(y), torch.full_like(y, 0.2)))
        return d

    def _emb_scale(self, ids, mask, seed, p, row_offset):
        rows = ids.numel()
        sc = torch.ones(rows, 1)
        if mask is not None:
            sc = mask.reshape(rows, 1).float()
        return sc, rows

    def emb_gather(self, W, ids, mask, seed, p, row_offset=0):
        E = W.shape[1]
        rows = ids.numel()
        out = W[ids.reshape(-1)]
        if mask is not None:
            out = out * mask.reshape(rows, 1)
        if p and p > 0:
            keep = rng.keep_mask(seed, rng.STREAM_EMB, np.arange(rows, dtype=np.uint32) + np.uint32(row_offset),
                                 np.arange(E, dtype=np.uint32), p)
            out = out * torch.from_numpy(keep.astype(np.float32) / np.float32(1 - p))
        return out.reshape(tuple(ids.shape) + (E,)).contiguous()

    def emb_scatter_add_(self, dW, ids, mask, dOut, seed, p, pad_idx=0, row_offset=0):
        E = dW.shape[1]
        rows = ids.numel()
        d = dOut.reshape(rows, E)
        if mask is not None:
            d = d * mask.reshape(rows, 1)
        if p and p > 0:
            keep = rng.keep_mask(seed, rng.STREAM_EMB, np.arange(rows, dtype=np.uint32) + np.uint32(row_offset),
                                 np.arange(E, dtype=np.uint32), p)
            d = d * torch.from_numpy(keep.astype(np.float32) / np.float32(1 - p))
        flat = ids.reshape(-1)
        sel = flat != pad_idx
        dW.index_add_(0, flat[sel], d[sel])
        return dW

    def conv_relu(self, e, Wc, bc):
        return torch.relu(F.conv1d(e.transpose(1, 2), Wc, bc)).transpose(1, 2).contiguous()

    def conv_bwd(self, dY, e, Wc, need_de=True):
        inp = e.transpose(1, 2)
        g = dY.transpose(1, 2)
        dWc = torch.nn.grad.conv1d_weight(inp, Wc.shape, g)
        dbc = dY.sum((0, 1))
        de = torch.nn.grad.conv1d_input(inp.shape, Wc, g).transpose(1, 2).contiguous() if need_de else None
        return de, dWc, dbc

    def label_avg(self, W, idx, mask):
        return ((W[idx] * mask[:, :, None]).sum(1) / mask.sum(1)[:, None]).contiguous()

    def label_avg_bwd_(self, dW, davg, idx, mask, pad_idx=0, inv=None):
        coef = mask / mask.sum(1, keepdim=True)
        contrib = (coef[:, :, None] * davg[:, None, :]).reshape(-1, davg.shape[1])
        flat = idx.reshape(-1)
        sel = flat != pad_idx
        dW.index_add_(0, flat[sel], contrib[sel])
        return dW

    def spmm(self, csr, x, pre=None, post=None, out=None, accumulate=False):
        rowptr, col, val = csr
        N = x.shape[0]
        rows = torch.repeat_interleave(torch.arange(N), rowptr[1:] - rowptr[:-1])
        w = val if pre is None else val * pre[col]
        r = torch.zeros_like(x).index_add_(0, rows, w[:, None] * x[col])
        if post is not None:
            r = r * post[:, None]
        if out is None:
            return r
        if accumulate:
            out += r
        else:
            out.copy_(r)
        return out

    def gru_gate_fwd(self, gi, gh, h):
        H = h.shape[1]
        r = torch.sigmoid(gi[:, :H] + gh[:, :H])
        z = torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
        n = torch.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:])
        return (1 - z) * n + z * h

    def gru_seq_step_(self, gi_all, t, gh, h, mask, pool, pool_mode):
        live = (t < mask.sum(1))[:, None]
        hn = self.gru_gate_fwd(gi_all[:, t], gh, h)
        h.copy_(torch.where(live, hn, h))
        if pool_mode == 'max':
            pool.copy_(torch.where(live & (hn != 0), torch.maximum(pool, hn), pool))
        elif pool_mode == 'mean':
            pool.add_(torch.where(live, hn, torch.zeros_like(hn)))
            if t == mask.shape[1] - 1:
                pool.div_(mask.sum(1)[:, None])

    def gru_gate_bwd(self, dhn, gi, gh, h):
        H = h.shape[1]
        r = torch.sigmoid(gi[:, :H] + gh[:, :H])
        z = torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
        ghn = gh[:, 2 * H:]
        n = torch.tanh(gi[:, 2 * H:] + r * ghn)
        dn_pre = dhn * (1 - z) * (1 - n * n)
        dz_pre = dhn * (h - n) * z * (1 - z)
        dr_pre = dn_pre * ghn * r * (1 - r)
        return (torch.cat([dr_pre, dz_pre, dn_pre], 1).contiguous(), torch.cat([dr_pre, dz_pre, dn_pre * r], 1).contiguous(),
                (dhn * z).contiguous())

    def _attn_drop(self, B, L, Tp, l_offset, L_total, seed, p):
        if not p or p <= 0:
            return None
        return torch.from_numpy(rng.attn_dropout_scale(seed, np.arange(B), l_offset + np.arange(L), L_total,
                                                       np.arange(Tp), p))

    def attn_fwd(self, PX, X, Q, l_offset, L_total, seed, p, out=None):
        S = torch.einsum('bte,le->blt', PX, Q)
        lse = torch.logsumexp(S, 2)
        P = torch.exp(S - lse[:, :, None])
        d = self._attn_drop(PX.shape[0], Q.shape[0], PX.shape[1], l_offset, L_total, seed, p)
        Pd = P if d is None else P * d
        if out is not None:
            out[0].copy_(Pd @ X)
            out[1].copy_(lse)
            return out
        return (Pd @ X).contiguous(), lse.contiguous()

    def attn_bwd(self, dO, O, PX, X, Q, lse, l_offset, L_total, seed, p, dPX, dX, dQ, nsplit=None, delta=None):
        S = torch.einsum('bte,le->blt', PX, Q)
        P = torch.exp(S - lse[:, :, None])
        d = self._attn_drop(PX.shape[0], Q.shape[0], PX.shape[1], l_offset, L_total, seed, p)
        Pd = P if d is None else P * d
        dPd = dO @ X.transpose(1, 2)
        dX += Pd.transpose(1, 2) @ dO
        dP = dPd if d is None else dPd * d
        delta = (dO * O).sum(2, keepdim=True) if delta is None else delta[:, :, None]
        dS = P * (dP - delta)
        dPX += torch.einsum('blt,le->bte', dS, Q)
        dQ += torch.einsum('blt,bte->le', dS, PX)

    def logits_fwd(self, direct, clf):
        return (torch.relu(direct) * clf[None]).sum(2).contiguous()

    def logits_bwd(self, direct, clf, dlogits, need_ddirect=True):
        dd = (dlogits[:, :, None] * clf[None] * (direct > 0).float()).contiguous() if need_ddirect else None
        dclf = torch.einsum('bl,bld->ld', dlogits, torch.relu(direct)).contiguous()
        return dd, dclf

    def logits_bwd_delta(self, direct, clf, bo, dlogits):
        dd, dclf = self.logits_bwd(direct, clf, dlogits)
        return dd, dclf, (dd * (direct - bo[None, None, :])).sum(2).contiguous()

    def ldam_bce(self, logits, y, margin, C, inv_b, need_grad=True, need_probs=False):
        z = logits
        if margin is not None and C > 0:
            z = logits - y * margin[None, :] * C
        loss = F.binary_cross_entropy_with_logits(z, y, reduction='sum') * inv_b
        dl = ((torch.sigmoid(z) - y) * inv_b).contiguous() if need_grad else None
        return loss, dl, (torch.sigmoid(logits) if need_probs else None)

    def multihot(self, y_indices, L):
        B, W = y_indices.shape
        y = torch.zeros(B, L)
        for b in range(B):
            row = y_indices[b]
            y[b, row[(row >= 0) & (row < L)]] = 1
        return y

    def eval_counts_(self, logits, y, counts, probs=None):
        Lc = counts.shape[1]
        p = torch.sigmoid(logits[:, :Lc])
        h, t = p > 0.5, y[:, :Lc] != 0
        counts += torch.stack([(h & t).sum(0), h.sum(0), t.sum(0), (h | t).sum(0)]).to(torch.int32)
        if probs is not None:
            probs.copy_(p)
        return counts

    def forward_final(self, fc, feats, labels, margin, C, scale, need_grad=True):
        rows = fc[labels]
        z = (rows * feats).sum(1)
        if margin is not None and C > 0:
            z = z - margin[labels] * C
        loss = F.binary_cross_entropy_with_logits(z, torch.ones_like(z), reduction='sum') * scale
        dfc = None
        if need_grad:
            dz = (torch.sigmoid(z) - 1) * scale
            dfc = torch.zeros_like(fc).index_add_(0, labels, dz[:, None] * feats)
        return loss, dfc

    def adamw_step_(self, p, g, m, v, lr, beta1, beta2, eps, wd):
        m.mul_(beta1).add_(g, alpha=1 - beta1)
        v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
        upd = m / (v.sqrt() + eps)
        if wd > 0:
            upd = upd + wd * p
        p.add_(upd, alpha=-lr)

    # ---- WGAN-GP pieces
    def gp_interp(self, real, fake, alpha):
        a = alpha.reshape(-1, 1)
        return a * real + (1 - a) * fake

    def gp_slope_v(self, h, w2):
        s = torch.where(h > 0, torch.ones_like(h), torch.full_like(h, 0.2))
        return s * w2[None, :], s

    def gp_rownorm_(self, g, lam, out=None):
        nrm = g.norm(2, dim=1)
        Bg = g.shape[0]
        gp = lam * ((nrm - 1) ** 2).mean()
        coef = lam * 2 * (nrm - 1) / Bg / torch.where(nrm > 0, nrm, torch.ones_like(nrm))
        g.mul_(coef[:, None])
        if out is None:
            return gp
        out += gp
        return out

    def gp_dw2_(self, dv, s, dw2):
        dw2 += (dv * s).sum(0)

    def rowdot(self, a, w, bias, lrelu_in=False):
        if lrelu_in:
            a = F.leaky_relu(a, 0.2)
        r = a @ w
        return r + bias[0] if bias is not None else r

    def weighted_sum(self, x, w, scale):
        return ((x * w).sum() if w is not None else x.sum()) * scale

    def rowdot_bwd(self, dout, w, act, dw, db, need_da=True, lrelu_in=False):
        if dw is not None:
            dw += dout @ (F.leaky_relu(act, 0.2) if lrelu_in else act)
        if db is not None:
            db += dout.sum()
        if need_da:
            return dout[:, None] * w[None, :] * torch.where(act > 0, torch.ones_like(act), torch.full_like(act, 0.2))
        return None

    def gan_critic_head(self, hf, lh, w2, sim, alpha, dw2, db1, cost):
        Bg, H = lh.shape
        wb = (torch.ones(Bg) if sim is None else sim) / Bg
        hr, hk = hf[:Bg] + lh, hf[Bg:] + lh
        a = alpha.reshape(-1, 1)
        slope = lambda x: torch.where(x > 0, torch.ones_like(x), torch.full_like(x, 0.2))  # noqa: E731
        hh = a * hr + (1 - a) * hk
        dr = -wb[:, None] * w2[None] * slope(hr)
        dk = wb[:, None] * w2[None] * slope(hk)
        diff = F.leaky_relu(hk, 0.2) - F.leaky_relu(hr, 0.2)
        dw2 += (wb[:, None] * diff).sum(0)
        cost += (wb[:, None] * diff * w2[None]).sum()
        db1 += (dr + dk).sum(0)
        return torch.cat([dr, dk, slope(hh) * w2[None]], 0).contiguous(), (dr + dk).contiguous(), slope(hh)

    def gan_gen_head(self, hf, lh, w2, b2, sim, cost):
        Bg, H = lh.shape
        wb = (torch.ones(Bg) if sim is None else sim) / Bg
        h = hf + lh
        cost -= (wb * (F.leaky_relu(h, 0.2) @ w2 + b2[0])).sum()
        return -wb[:, None] * w2[None] * torch.where(h > 0, torch.ones_like(h), torch.full_like(h, 0.2))

    ADAM_STATE_INTS = 1024

    def adam_flat_step_(self, p, g, m, v, state, lr, beta1, beta2, eps):
        t = int(state[0]) + 1
        state += 1
        m.add_((g - m) * (1 - beta1))
        v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
        p.sub_((lr / (1 - beta1 ** t)) * m / (v.sqrt() / (1 - beta2 ** t) ** 0.5 + eps))

    def keyword_ce_sim_(self, logits, kw_idx, kw_mask, sim, grad_scale, loss_scale, loss_sum):
        Bg = logits.shape[0]
        kww = torch.ones(Bg) if sim is None else (sim == 1).float()
        loss_b = self.keyword_ce_(logits, kw_idx, kw_mask, kww * grad_scale)
        if loss_sum is not None:
            loss_sum += (loss_b * kww).sum() * loss_scale
        return loss_b

    def keyword_ce_(self, logits, kw_idx, kw_mask, w):
        Bg, Vk = logits.shape
        nm = kw_mask / (kw_mask.sum(1, keepdim=True) + 1e-7)
        t = torch.zeros(Bg, Vk)
        for j in range(kw_idx.shape[1]):
            t[torch.arange(Bg), kw_idx[:, j]] = nm[:, j]
        logp = torch.log_softmax(logits, 1)
        loss_b = -(logp * t).sum(1)
        wb = w if w is not None else torch.ones(Bg)
        logits.copy_(wb[:, None] * (torch.exp(logp) * t.sum(1, keepdim=True) - t))
        return loss_b
