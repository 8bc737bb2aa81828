"""numpy restatement of the base feature extractor (forward + hand-derived backward).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Follows, function by function:

  average_label_embedding   /root/reference/modules/icd_modules.py:110-113
  GraphGatedEncoder.forward /root/reference/modules/graph_module.py:104-129   (2-D branch)
  ConvGraphEncoder.forward  /root/reference/modules/graph_module.py:32-81     (2-D branch)
  forward_direct_feats      /root/reference/modules/icd_modules.py:152-160
  conv_attn                 /root/reference/modules/icd_modules.py:125-150
  forward / forward_loss    /root/reference/modules/icd_modules.py:162-186
  ConvLabelAttnGAN.forward  /root/reference/modules/icd_modules.py:251-282
  forward_final             /root/reference/modules/icd_modules.py:233-249
  BCEWithLogitsLoss(sum)    /root/reference/modules/common_module.py:29-30
  AdamW.step                /root/reference/modules/common_module.py:138-182

Parity pin: tests/test_oracle_golden.py checks every function here against
tests/golden/*.npz, which tests/golden/make_golden.py captured from the reference
modules imported on CPU.  Dropout uses the build's own counter RNG (oracle/rng.py);
train-mode goldens were produced by injecting that same mask into the reference.

Parameter names are the reference's state-dict keys.
"""
import numpy as np
from numpy.lib.stride_tricks import sliding_window_view

from . import rng

EPS_DEG = 1e-7  # graph_module.py:117


def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


# ----------------------------------------------------------------------------------------------
# label side
# ----------------------------------------------------------------------------------------------
def avg_label_emb_fwd(W, idx, mask):
    """icd_modules.py:110-113: sum_j mask[n,j] * W[idx[n,j]] / sum_j mask[n,j]."""
    g = W[idx]  # [N, Td, E]
    s = (g * mask[:, :, None]).sum(1)
    return s / mask.sum(1)[:, None]


def avg_label_emb_bwd(dW, davg, idx, mask, pad_idx=0):
    """Scatter-add into the dense embedding grad; nn.Embedding(padding_idx=0) drops row 0."""
    coef = mask / mask.sum(1)[:, None]  # [N, Td]
    contrib = coef[:, :, None] * davg[:, None, :]
    flat_idx = idx.reshape(-1)
    keep = flat_idx != pad_idx
    np.add.at(dW, flat_idx[keep], contrib.reshape(-1, davg.shape[1])[keep])


def adj_to_coo(adj):
    """Accepts a dense [N,N] array or a (rows, cols, vals, N) tuple; returns the tuple."""
    if isinstance(adj, tuple):
        return adj
    r, c = np.nonzero(adj)
    return r.astype(np.int64), c.astype(np.int64), adj[r, c].astype(adj.dtype), adj.shape[0]


def spmm(coo, h, transpose=False):
    r, c, v, n = coo
    if transpose:
        r, c = c, r
    out = np.zeros((n, h.shape[1]), h.dtype)
    np.add.at(out, r, v[:, None].astype(h.dtype) * h[c])
    return out


def gru_cell_fwd(u, h, Wih, Whh, bih, bhh):
    H = h.shape[1]
    gi = u @ Wih.T + bih
    gh = h @ Whh.T + bhh
    r = _sigmoid(gi[:, :H] + gh[:, :H])
    z = _sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
    n = np.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:])
    hn = (1.0 - z) * n + z * h
    return hn, (u, h, r, z, n, gh[:, 2 * H:])


def gru_cell_bwd(dhn, cache, Wih, Whh, grads, prefix):
    u, h, r, z, n, ghn = cache
    dn = dhn * (1.0 - z)
    dz = dhn * (h - n)
    dh = dhn * z
    dn_pre = dn * (1.0 - n * n)
    dr = dn_pre * ghn
    dr_pre = dr * r * (1.0 - r)
    dz_pre = dz * z * (1.0 - z)
    dgi = np.concatenate([dr_pre, dz_pre, dn_pre], 1)
    dgh = np.concatenate([dr_pre, dz_pre, dn_pre * r], 1)
    grads[prefix + 'weight_ih'] += dgi.T @ u
    grads[prefix + 'weight_hh'] += dgh.T @ h
    grads[prefix + 'bias_ih'] += dgi.sum(0)
    grads[prefix + 'bias_hh'] += dgh.sum(0)
    du = dgi @ Wih
    dh = dh + dgh @ Whh
    return du, dh


def gate_encoder_fwd(x, coo, deg, p, steps=2):
    """graph_module.py:104-129."""
    pre = 'graph_label_encoder.gru.'
    inv = 1.0 / (deg + x.dtype.type(EPS_DEG))
    h = x
    caches = []
    for _ in range(steps):
        u = spmm(coo, h) * inv[:, None]
        h, c = gru_cell_fwd(u, h, p[pre + 'weight_ih'], p[pre + 'weight_hh'], p[pre + 'bias_ih'], p[pre + 'bias_hh'])
        caches.append(c)
    return np.concatenate([x, h], 1), (caches, inv)


def gate_encoder_bwd(dout, cache, coo, p, grads, E):
    pre = 'graph_label_encoder.gru.'
    caches, inv = cache
    dx = dout[:, :E].copy()
    dh = dout[:, E:].copy()
    for c in reversed(caches):
        du, dh = gru_cell_bwd(dh, c, p[pre + 'weight_ih'], p[pre + 'weight_hh'], grads, pre)
        dh = dh + spmm(coo, du * inv[:, None], transpose=True)
    return dx + dh


def conv_encoder_fwd(x, coo, deg, p):
    """graph_module.py:32-81 (ConvGraphEncoder, 2-D branch)."""
    inv = 1.0 / (deg + x.dtype.type(EPS_DEG))
    caches = []
    f = x
    for name in ('gc1', 'gc2'):
        W = p[f'graph_label_encoder.{name}.fc.weight']
        b = p[f'graph_label_encoder.{name}.fc.bias']
        nb = spmm(coo, f) * inv[:, None]
        cat = np.concatenate([f, nb], 1)
        out = np.maximum(cat @ W.T + b, 0)
        caches.append((cat, out, f.shape[1]))
        f = out
    return np.concatenate([f, x], 1), (caches, inv)


def conv_encoder_bwd(dout, cache, coo, p, grads, H):
    caches, inv = cache
    df = dout[:, :H].copy()
    dx = dout[:, H:].copy()
    for name, (cat, out, din) in zip(('gc2', 'gc1'), reversed(caches)):
        W = p[f'graph_label_encoder.{name}.fc.weight']
        dpre = df * (out > 0)
        grads[f'graph_label_encoder.{name}.fc.weight'] += dpre.T @ cat
        grads[f'graph_label_encoder.{name}.fc.bias'] += dpre.sum(0)
        dcat = dpre @ W
        df = dcat[:, :din] + spmm(coo, dcat[:, din:] * inv[:, None], transpose=True)
    return dx + df


def label_embedding_fwd(p, bufs, graph_encoder='gate'):
    """icd_modules.py:115-123 -> (avg [N,E], graph [N,D], cache)."""
    W = p['emb.embed.weight']
    avg = avg_label_emb_fwd(W, bufs['code_idx_matrix'], bufs['code_idx_mask'])
    coo = adj_to_coo(bufs['adj_matrix'])
    if graph_encoder == 'gate':
        graph, c = gate_encoder_fwd(avg, coo, bufs['num_neighbors'], p)
    elif graph_encoder == 'conv':
        graph, c = conv_encoder_fwd(avg, coo, bufs['num_neighbors'], p)
    else:
        graph, c = avg, None
    return avg, graph, (coo, c)


def label_embedding_bwd(davg, dgraph, cache, p, bufs, grads, graph_encoder='gate'):
    coo, c = cache
    E = p['emb.embed.weight'].shape[1]
    if graph_encoder == 'gate':
        davg = davg + gate_encoder_bwd(dgraph, c, coo, p, grads, E)
    elif graph_encoder == 'conv':
        davg = davg + conv_encoder_bwd(dgraph, c, coo, p, grads, dgraph.shape[1] - E)
    else:
        davg = davg + dgraph
    avg_label_emb_bwd(grads['emb.embed.weight'], davg, bufs['code_idx_matrix'], bufs['code_idx_mask'])


# ----------------------------------------------------------------------------------------------
# note encoder
# ----------------------------------------------------------------------------------------------
def encoder_fwd(p, x, mask, drop_seed=None, p_emb=0.5):
    """emb gather * mask * dropout -> conv1d+relu -> proj+tanh.  icd_modules.py:153-157,130-135."""
    W = p['emb.embed.weight']
    B, T = x.shape
    E = W.shape[1]
    e = W[x]
    scale = np.ones((B, T, 1), W.dtype)
    if mask is not None:
        scale = mask[:, :, None].astype(W.dtype)
    if drop_seed is not None:
        scale = scale * rng.emb_dropout_scale(drop_seed, B, T, E, p_emb).astype(W.dtype)
    e = e * scale
    Wc = p['conv_modules.0.weight']  # [F, E, k]
    F_, _, k = Wc.shape
    win = sliding_window_view(e, k, axis=1)  # [B, T', E, k]
    Tp = win.shape[1]
    A = win.reshape(B * Tp, E * k)
    X = np.maximum(A @ Wc.reshape(F_, E * k).T + p['conv_modules.0.bias'], 0).reshape(B, Tp, F_)
    PX = np.tanh(X @ p['proj.weight'].T + p['proj.bias'])
    return X, PX, (x, scale, e, A)


def encoder_bwd(dX, dPX, X, PX, cache, p, grads, pad_idx=0):
    x, scale, e, A = cache
    B, Tp, F_ = X.shape
    Wc = p['conv_modules.0.weight']
    _, E, k = Wc.shape
    dpre = dPX * (1.0 - PX * PX)
    grads['proj.weight'] += dpre.reshape(-1, dpre.shape[2]).T @ X.reshape(-1, F_)
    grads['proj.bias'] += dpre.sum((0, 1))
    dX = dX + dpre @ p['proj.weight']
    dY = (dX * (X > 0)).reshape(B * Tp, F_)
    grads['conv_modules.0.weight'] += (dY.T @ A).reshape(F_, E, k)
    grads['conv_modules.0.bias'] += dY.sum(0)
    dA = (dY @ Wc.reshape(F_, E * k)).reshape(B, Tp, E, k)
    de = np.zeros_like(e)
    for j in range(k):
        de[:, j:j + Tp, :] += dA[:, :, :, j]
    de = de * scale
    flat = x.reshape(-1)
    keep = flat != pad_idx
    np.add.at(grads['emb.embed.weight'], flat[keep], de.reshape(-1, E)[keep])


# ----------------------------------------------------------------------------------------------
# label attention + classifier + loss
# ----------------------------------------------------------------------------------------------
def label_attn_fwd(PX, X, Q, drop_seed=None, p_attn=0.2, l_offset=0, L_total=None):
    """softmax over tokens, dropout on probabilities, pooling.  icd_modules.py:137-142."""
    B, Tp, _ = PX.shape
    L = Q.shape[0]
    L_total = L if L_total is None else L_total
    S = np.einsum('bte,le->blt', PX, Q, optimize=True)
    m = S.max(2, keepdims=True)
    Pn = np.exp(S - m)
    den = Pn.sum(2, keepdims=True)
    P = Pn / den
    lse = (m + np.log(den))[:, :, 0]
    if drop_seed is not None:
        drop = rng.attn_dropout_scale(drop_seed, np.arange(B), l_offset + np.arange(L), L_total,
                                      np.arange(Tp), p_attn).astype(P.dtype)
        Pd = P * drop
    else:
        drop = None
        Pd = P
    O = Pd @ X
    return O, lse, (P, drop)


def label_attn_bwd(dO, PX, X, Q, cache):
    P, drop = cache
    Pd = P if drop is None else P * drop
    dPd = dO @ X.transpose(0, 2, 1)
    dX = Pd.transpose(0, 2, 1) @ dO
    dP = dPd if drop is None else dPd * drop
    dS = P * (dP - (P * dP).sum(2, keepdims=True))
    dPX = np.einsum('blt,le->bte', dS, Q, optimize=True)
    dQ = np.einsum('blt,bte->le', dS, PX, optimize=True)
    return dPX, dX, dQ


def bce_with_logits_sum(z, y):
    """common_module.py:29-30 (reduction='sum')."""
    return (np.maximum(z, 0) - z * y + np.log1p(np.exp(-np.abs(z)))).sum()


def classifier_loss_fwd(O, Wo, bo, clf_rows, y, margin, C):
    """output_proj + relu + per-label dot + LDAM/BCE.  icd_modules.py:159,177-185,162-172."""
    direct = O @ Wo.T + bo
    feat = np.maximum(direct, 0)
    logits = (feat * clf_rows[None]).sum(2)
    z = logits
    if margin is not None and C > 0:
        z = logits - y * margin[None, :] * logits.dtype.type(C)
    loss = bce_with_logits_sum(z, y) / y.shape[0]
    return logits, loss, (direct, feat, z)


def classifier_loss_bwd(O, Wo, clf_rows, y, cache):
    direct, feat, z = cache
    B = y.shape[0]
    dlogits = (_sigmoid(z) - y) / B
    dclf = np.einsum('bl,bld->ld', dlogits, feat, optimize=True)
    ddirect = dlogits[:, :, None] * clf_rows[None] * (direct > 0)
    dWo = ddirect.reshape(-1, ddirect.shape[2]).T @ O.reshape(-1, O.shape[2])
    dbo = ddirect.sum((0, 1))
    dO = ddirect @ Wo
    return dO, dWo, dbo, dclf, dlogits


def zero_grads(p):
    return {k: np.zeros_like(v) for k, v in p.items()}


def base_forward_backward(p, bufs, x, y, mask, L, margin, C, graph_encoder='gate', drop_seed=None,
                          need_grad=True, l_chunk=None):
    """ConvLabelAttnModel.forward + loss.backward() (icd_modules.py:174-186; train_base.py:109-112).

    Returns dict(logits, loss, grads, avg, graph, direct, attn).  `l_chunk` processes the label
    dimension in slices to bound the [B, l, T'] buffers (the reference materialises them whole).
    """
    dt = p['emb.embed.weight'].dtype
    avg, graph, lcache = label_embedding_fwd(p, bufs, graph_encoder)
    X, PX, ecache = encoder_fwd(p, x, mask, drop_seed)
    Wo, bo = p['output_proj.weight'], p['output_proj.bias']
    B = x.shape[0]
    step = L if l_chunk is None else l_chunk
    logits = np.zeros((B, L), dt)
    loss = dt.type(0)
    grads = zero_grads(p) if need_grad else None
    davg = np.zeros_like(avg)
    dgraph = np.zeros_like(graph)
    dX = np.zeros_like(X)
    dPX = np.zeros_like(PX)
    keep = {}
    for l0 in range(0, L, step):
        l1 = min(L, l0 + step)
        Q = avg[l0:l1]
        O, lse, acache = label_attn_fwd(PX, X, Q, drop_seed, l_offset=l0, L_total=L)
        lg, ls, ccache = classifier_loss_fwd(O, Wo, bo, graph[l0:l1], y[:, l0:l1],
                                             None if margin is None else margin[l0:l1], C)
        logits[:, l0:l1] = lg
        loss = loss + ls
        if l_chunk is None:
            keep = dict(direct=ccache[0], attn=O)
        if need_grad:
            dO, dWo, dbo, dclf, _ = classifier_loss_bwd(O, Wo, graph[l0:l1], y[:, l0:l1], ccache)
            grads['output_proj.weight'] += dWo
            grads['output_proj.bias'] += dbo
            dgraph[l0:l1] += dclf
            dpx, dx_, dq = label_attn_bwd(dO, PX, X, Q, acache)
            dPX += dpx
            dX += dx_
            davg[l0:l1] += dq
    if need_grad:
        encoder_bwd(dX, dPX, X, PX, ecache, p, grads)
        label_embedding_bwd(davg, dgraph, lcache, p, bufs, grads, graph_encoder)
    out = dict(logits=logits, loss=loss, grads=grads, avg=avg, graph=graph, X=X, PX=PX)
    out.update(keep)
    return out


def gan_model_forward_backward(p, frozen_avg, x, y, mask, L, margin, C, label_indices=None, need_grad=True,
                               drop_seed=None):
    """ConvLabelAttnGAN.forward (icd_modules.py:251-282), only output_fc trainable.  drop_seed=None: eval mode;
    otherwise train mode as finetune.py:258,271 runs it -- embedding dropout (icd_modules.py:157) and attention dropout
    (:140) from the counter RNG, the attention mask keyed on the rows of the ATTENDED list (a gathered subset is numbered
    0 .. |sub|-1: the reference draws a fresh [B, |sub|, T'] mask there)."""
    fc = p['output_fc.weight']
    Q = frozen_avg
    if label_indices is not None:
        Q = Q[label_indices]
        y = y[:, label_indices]
        margin = None if margin is None else margin[label_indices]
        rows = fc[label_indices]
    else:
        rows = fc
    X, PX, _ = encoder_fwd(p, x, mask, drop_seed)
    Q = Q[:L]
    rows = rows[:L]
    O, _, _ = label_attn_fwd(PX, X, Q, drop_seed, l_offset=0, L_total=Q.shape[0])
    logits, loss, cc = classifier_loss_fwd(O, p['output_proj.weight'], p['output_proj.bias'], rows, y,
                                           None if margin is None else margin[:L], C)
    dfc = None
    if need_grad:
        _, _, _, dclf, _ = classifier_loss_bwd(O, p['output_proj.weight'], rows, y, cc)
        dfc = np.zeros_like(fc)
        if label_indices is not None:
            np.add.at(dfc, np.asarray(label_indices)[:L], dclf)
        else:
            dfc[:L] += dclf
    return dict(logits=logits, loss=loss, dfc=dfc, direct=cc[0], attn=O)


def forward_final(fc, feats, labels, margin, C, need_grad=True):
    """icd_modules.py:233-249: gather classifier rows, row-dot, LDAM shift, BCE(sum) vs ones."""
    rows = fc[labels]
    logits = (rows * feats).sum(1)
    z = logits
    if margin is not None and C > 0:
        z = logits - margin[labels] * logits.dtype.type(C)
    y = np.ones_like(z)
    loss = bce_with_logits_sum(z, y)
    dfc = None
    dfeats = None
    if need_grad:
        dz = _sigmoid(z) - y
        dfc = np.zeros_like(fc)
        np.add.at(dfc, labels, dz[:, None] * feats)
        dfeats = dz[:, None] * rows
    return loss, dfc, dfeats


def adamw_nobias_step(pv, g, m, v, lr, beta1, beta2, eps, wd):
    """common_module.py:138-182: no bias correction, eps added to sqrt(v), decoupled wd on p."""
    dt = pv.dtype.type
    m = m * dt(beta1) + dt(1 - beta1) * g
    v = v * dt(beta2) + dt(1 - beta2) * g * g
    upd = m / (np.sqrt(v) + dt(eps))
    if wd > 0:
        upd = upd + dt(wd) * pv
    return pv - dt(lr) * upd, m, v


def class_margin_from_count(target_count):
    """icd_modules.py:80-83."""
    cm = np.asarray(target_count, np.float32) ** np.float32(0.25)
    cm = np.where(cm == 0, np.float32(1), cm)
    return (np.float32(1.0) / cm).astype(np.float32)
