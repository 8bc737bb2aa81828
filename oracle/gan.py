"""numpy restatement of the WGAN-GP feature generator / critic step.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Follows:

  ConditionalGenerator.forward      /root/reference/modules/generative_module.py:41-53
  ConditionalDiscriminator.forward  /root/reference/modules/generative_module.py:56-68
  calc_gradient_penalty             /root/reference/modules/generative_module.py:22-38
        restated in closed form: for a 2-layer LeakyReLU critic the input gradient is
        g = W1[:, :D]^T (s * w2), s = lrelu'(W1 [x^; lab] + b1) in {1, 0.2}; the penalty's
        parameter gradient reaches only W1[:, :D] and w2 (s is piecewise constant).
  LinearKeywordsDecoder.forward     /root/reference/modules/decoder_module.py:39-59
  RNNLabelEncoder.forward_enc       /root/reference/modules/lm_module.py:68-88  (pool 'max'/'mean')
  one WGAN-GP step (5 critic : 1 generator)   /root/reference/train_gan.py:223-308
  torch.optim.Adam (betas .5/.999, bias-corrected)   as called at train_gan.py:164-166

Pinned by tests/test_oracle_golden.py against tests/golden/gan_*.npz captured from the reference.
"""
import numpy as np

LRELU = 0.2


def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def _lrelu(x):
    return np.where(x > 0, x, x.dtype.type(LRELU) * x)


def _slope(x):
    return np.where(x > 0, x.dtype.type(1), x.dtype.type(LRELU))


def mlp2_fwd(a, b, W1, b1, W2, b2):
    """cat[a, b] -> Linear -> LeakyReLU(0.2) -> Linear."""
    inp = np.concatenate([a, b], 1)
    h = inp @ W1.T + b1
    act = _lrelu(h)
    out = act @ W2.T + b2
    return out, (inp, h, act)


def mlp2_bwd(dout, cache, W1, W2, n_first):
    """Returns grads (dW1, db1, dW2, db2) and the gradient w.r.t. the first n_first input columns."""
    inp, h, act = cache
    dW2 = dout.T @ act
    db2 = dout.sum(0)
    dact = dout @ W2
    dh = dact * _slope(h)
    dW1 = dh.T @ inp
    db1 = dh.sum(0)
    dfirst = dh @ W1[:, :n_first]
    return (dW1, db1, dW2, db2), dfirst


def critic_fwd(feat, lab, d):
    out, c = mlp2_fwd(feat, lab, d['fc1.weight'], d['fc1.bias'], d['fc2.weight'], d['fc2.bias'])
    return out[:, 0], c


def gradient_penalty(real, fake, lab, alpha, d, lam=10.0):
    """Closed-form value and parameter gradients of generative_module.py:22-38."""
    W1, b1, w2 = d['fc1.weight'], d['fc1.bias'], d['fc2.weight'][0]
    D = real.shape[1]
    Bg = real.shape[0]
    xhat = alpha * real + (1 - alpha) * fake
    h = np.concatenate([xhat, lab], 1) @ W1.T + b1
    s = _slope(h)
    v = s * w2[None, :]
    W1f = W1[:, :D]
    g = v @ W1f
    nrm = np.sqrt((g * g).sum(1))
    dt = real.dtype.type
    gp = dt(lam) * ((nrm - 1) ** 2).mean()
    coef = dt(lam) * 2 * (nrm - 1) / Bg / np.where(nrm > 0, nrm, 1)
    Gp = coef[:, None] * g
    dv = Gp @ W1f.T
    dW1 = np.zeros_like(W1)
    dW1[:, :D] = v.T @ Gp
    dw2 = (dv * s).sum(0)
    grads = {'fc1.weight': dW1, 'fc1.bias': np.zeros_like(b1), 'fc2.weight': dw2[None, :],
             'fc2.bias': np.zeros_like(d['fc2.bias'])}
    return gp, grads


def keyword_targets(kw_idx, kw_mask, Vk):
    """decoder_module.py:46-48,55: normalised mask scattered by assignment (last duplicate wins)."""
    nm = kw_mask / (kw_mask.sum(1, keepdims=True) + kw_mask.dtype.type(1e-7))
    t = np.zeros((kw_idx.shape[0], Vk), kw_mask.dtype)
    for j in range(kw_idx.shape[1]):
        t[np.arange(kw_idx.shape[0]), kw_idx[:, j]] = nm[:, j]
    return t


def keyword_fwd(feat, kw_idx, kw_mask, k):
    xk = feat @ k['proj_fc.weight'].T + k['proj_fc.bias']
    logits = xk @ k['word_emb'].T
    m = logits.max(1, keepdims=True)
    lse = m + np.log(np.exp(logits - m).sum(1, keepdims=True))
    logp = logits - lse
    t = keyword_targets(kw_idx, kw_mask, k['word_emb'].shape[0])
    loss = -(logp * t).sum(1)
    return loss, (xk, logp, t)


def keyword_bwd(dloss, feat, cache, k):
    xk, logp, t = cache
    dlogits = dloss[:, None] * (np.exp(logp) * t.sum(1, keepdims=True) - t)
    dxk = dlogits @ k['word_emb']
    dW = dxk.T @ feat
    db = dxk.sum(0)
    dfeat = dxk @ k['proj_fc.weight']
    return {'proj_fc.weight': dW, 'proj_fc.bias': db}, dfeat


def gru_label_encoder(word_emb, ids, mask, r, pool_mode='max'):
    """lm_module.py:68-88 with training=False.  r holds rnn.weight_ih_l0 etc.  Returns [n, E]."""
    dt = word_emb.dtype
    x = word_emb[ids] * mask[:, :, None]
    n, Td, E = x.shape
    lengths = mask.sum(1).astype(np.int64)
    Tmax = int(lengths.max())
    h = np.zeros((n, E), dt)
    outs = np.zeros((n, Tmax, E), dt)
    Wih, Whh, bih, bhh = r['rnn.weight_ih_l0'], r['rnn.weight_hh_l0'], r['rnn.bias_ih_l0'], r['rnn.bias_hh_l0']
    for t in range(Tmax):
        gi = x[:, t] @ Wih.T + bih
        gh = h @ Whh.T + bhh
        rr = _sigmoid(gi[:, :E] + gh[:, :E])
        zz = _sigmoid(gi[:, E:2 * E] + gh[:, E:2 * E])
        nn_ = np.tanh(gi[:, 2 * E:] + rr * gh[:, 2 * E:])
        hn = (1 - zz) * nn_ + zz * h
        live = (t < lengths)[:, None]
        h = np.where(live, hn, h)
        outs[:, t] = np.where(live, hn, 0)
    if pool_mode == 'mean':
        return outs.sum(1) / lengths[:, None].astype(dt)
    if pool_mode == 'max':
        return np.where(outs == 0, -np.inf, outs).max(1).astype(dt)
    return h


def adam_step(p, g, state, lr, beta1=0.5, beta2=0.999, eps=1e-8):
    """torch.optim.Adam, no weight decay, as constructed at train_gan.py:164-166."""
    dt = p.dtype.type
    state['t'] = state.get('t', 0) + 1
    t = state['t']
    m = state.get('m', np.zeros_like(p)) * dt(beta1) + dt(1 - beta1) * g
    v = state.get('v', np.zeros_like(p)) * dt(beta2) + dt(1 - beta2) * g * g
    state['m'], state['v'] = m, v
    bc1 = 1 - beta1 ** t
    bc2 = 1 - beta2 ** t
    denom = np.sqrt(v) / dt(np.sqrt(bc2)) + dt(eps)
    return p - dt(lr / bc1) * (m / denom)


def critic_iteration(d, g, real, lab, noise, alpha, sim, lam=10.0):
    """train_gan.py:228-260 up to (not including) optimizer_d.step().  Returns (d_cost, gp, grads)."""
    Bg, D = real.shape
    dt = real.dtype.type
    w = (np.ones(Bg, real.dtype) if sim is None else sim) / dt(Bg)
    real_logits, c_real = critic_fwd(real, lab, d)
    critic_real = (real_logits * w).sum()
    (dW1, db1, dW2, db2), _ = mlp2_bwd(-w[:, None], c_real, d['fc1.weight'], d['fc2.weight'], D)
    fake_pre, _ = mlp2_fwd(noise, lab, g['fc1.weight'], g['fc1.bias'], g['fc2.weight'], g['fc2.bias'])
    fake = np.maximum(fake_pre, 0)
    fake_logits, c_fake = critic_fwd(fake, lab, d)
    critic_fake = (fake_logits * w).sum()
    (eW1, eb1, eW2, eb2), _ = mlp2_bwd(w[:, None], c_fake, d['fc1.weight'], d['fc2.weight'], D)
    gp, ggp = gradient_penalty(real, fake, lab, alpha, d, lam)
    grads = {'fc1.weight': dW1 + eW1 + ggp['fc1.weight'], 'fc1.bias': db1 + eb1,
             'fc2.weight': dW2 + eW2 + ggp['fc2.weight'], 'fc2.bias': db2 + eb2}
    return critic_fake - critic_real, gp, grads, fake


def generator_iteration(d, g, k, lab, noise, sim, kw_idx, kw_mask, reg_ratio):
    """train_gan.py:262-306 up to (not including) optimizer_g.step()."""
    Bg = lab.shape[0]
    dt = lab.dtype.type
    w = (np.ones(Bg, lab.dtype) if sim is None else sim) / dt(Bg)
    fake_pre, c_g = mlp2_fwd(noise, lab, g['fc1.weight'], g['fc1.bias'], g['fc2.weight'], g['fc2.bias'])
    fake = np.maximum(fake_pre, 0)
    D = fake.shape[1]
    dfake = np.zeros_like(fake)
    kgrads = None
    kw_loss = dt(0)
    if reg_ratio > 0:
        kl, c_k = keyword_fwd(fake, kw_idx, kw_mask, k)
        kww = np.ones(Bg, lab.dtype) if sim is None else np.where(sim != 1, dt(0), sim)
        kw_loss = (kl * kww).mean()
        kgrads, dfk = keyword_bwd(dt(reg_ratio) * kww / dt(Bg), fake, c_k, k)
        dfake += dfk
    fake_logits, c_d = critic_fwd(fake, lab, d)
    g_cost = -(fake_logits * w).sum()
    _, dfd = mlp2_bwd(-w[:, None], c_d, d['fc1.weight'], d['fc2.weight'], D)
    dfake += dfd
    dpre = dfake * (fake_pre > 0)
    (gW1, gb1, gW2, gb2), _ = mlp2_bwd(dpre, c_g, g['fc1.weight'], g['fc2.weight'], 0)
    ggrads = {'fc1.weight': gW1, 'fc1.bias': gb1, 'fc2.weight': gW2, 'fc2.bias': gb2}
    return g_cost, kw_loss, ggrads, kgrads, fake
