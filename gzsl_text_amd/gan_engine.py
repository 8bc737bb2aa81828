"""WGAN-GP feature generator / critic step on HIP kernels (the hot loop of train_gan.py:223-308).

Parameters are dicts keyed like the reference's state dicts:
  generator / discriminator: fc1.weight [h, a+b], fc1.bias, fc2.weight [o, h], fc2.bias   (generative_module.py:41-68)
  keyword head: proj_fc.weight [E, D], proj_fc.bias, word_emb [Vk, E] (frozen buffer)       (decoder_module.py:25-31)

What differs from "run the reference's ops one by one":
  * torch.cat([x, label]) is never formed: fc1 is two GEMMs over the column blocks of fc1.weight, the second
    accumulating into the first's output with the activation in its epilogue.
  * The label half of the critic's fc1 (lab @ W1[:, D:]^T + b1) is shared by D(real), D(fake) and the penalty.
  * The gradient penalty uses the closed form (no double backward), and its fc1 pre-activation is the alpha-blend of
    the real / fake pre-activations (fc1 is affine and the label half is common), so it costs no extra fc1 GEMM.
  * The critic's single-output fc2 is a fused row-dot with LeakyReLU applied on the fly.
Sign conventions follow the driver: critic grads = d(-critic_real + critic_fake + gp), generator grads =
d(-critic_fake + reg_ratio * keyword_loss) (train_gan.py:244-256, 300-306).
"""
import torch

from .hip import ACT_LRELU, ACT_NONE, ACT_RELU


def zero_grads(params, skip=()):
    """Zero gradient tensors for a parameter dict, carved out of ONE flat buffer (one fill launch instead of one per tensor)."""
    keys = [k for k in params if k not in skip]
    flat = torch.zeros(sum(params[k].numel() for k in keys), dtype=torch.float32, device=params[keys[0]].device)
    out, off = {}, 0
    for k in keys:
        n = params[k].numel()
        out[k] = flat[off:off + n].view_as(params[k])
        off += n
    return out


class GanEngine:
    def __init__(self, ops):
        self.ops = ops

    # ------------------------------------------------------------------------------------------ generator
    def generator_fwd(self, g, noise, lab, relu_out=True):
        """relu(G(noise, lab)) (generative_module.py:48-53 + train_gan.py:248).  Returns (fake, hidden activations)."""
        ops = self.ops
        n1 = noise.shape[1]
        W1 = g['fc1.weight']
        h = ops.mm(noise, W1[:, :n1].t(), bias=g['fc1.bias'])
        ops.mm(lab, W1[:, n1:].t(), out=h, accumulate=True, act=ACT_LRELU)
        fake = ops.mm(h, g['fc2.weight'].t(), bias=g['fc2.bias'], act=ACT_RELU if relu_out else ACT_NONE)
        return fake, h

    def generator_bwd(self, dfake, fake, h, g, noise, lab, G, relu_out=True):
        """dfake is consumed.  Accumulates generator parameter gradients into G."""
        ops = self.ops
        n1 = noise.shape[1]
        if relu_out:
            ops.act_bwd_(dfake, fake, ACT_RELU)
        ops.mm(dfake.t(), h, out=G['fc2.weight'], accumulate=True)
        ops.colsum(dfake, out=G['fc2.bias'])
        dh = ops.mm(dfake, g['fc2.weight'])
        ops.act_bwd_(dh, h, ACT_LRELU)
        ops.mm(dh.t(), noise, out=G['fc1.weight'][:, :n1], accumulate=True)
        ops.mm(dh.t(), lab, out=G['fc1.weight'][:, n1:], accumulate=True)
        ops.colsum(dh, out=G['fc1.bias'])

    # ------------------------------------------------------------------------------------------ critic
    def critic_label_half(self, d, lab, D):
        """lab @ W1[:, D:]^T + b1, shared by every critic evaluation on the same labels."""
        return self.ops.mm(lab, d['fc1.weight'][:, D:].t(), bias=d['fc1.bias'])

    def critic_fwd(self, d, feat, lab_half):
        """Returns (logits [Bg], fc1 pre-activations [Bg, ndh])."""
        ops = self.ops
        D = feat.shape[1]
        h = lab_half.clone()
        ops.mm(feat, d['fc1.weight'][:, :D].t(), out=h, accumulate=True)
        logits = ops.rowdot(h, d['fc2.weight'][0], d['fc2.bias'], lrelu_in=True)
        return logits, h

    def critic_bwd(self, dout, h, d, G=None, need_dh=True):
        """dout [Bg] = d loss / d logits.  Adds fc2 grads into G (when given); returns dh = d loss / d pre-activations."""
        return self.ops.rowdot_bwd(dout, d['fc2.weight'][0], h, None if G is None else G['fc2.weight'][0],
                                   None if G is None else G['fc2.bias'], need_da=need_dh, lrelu_in=True)

    def critic_fc1_grads(self, dh, feat, lab, G):
        ops = self.ops
        D = feat.shape[1]
        ops.mm(dh.t(), feat, out=G['fc1.weight'][:, :D], accumulate=True)
        if lab is not None:
            ops.mm(dh.t(), lab, out=G['fc1.weight'][:, D:], accumulate=True)
            ops.colsum(dh, out=G['fc1.bias'])

    def gradient_penalty(self, d, h_real, h_fake, alpha, D, G, lam=10.0):
        """Closed-form generative_module.py:22-38 from the real / fake fc1 pre-activations.  Adds its parameter
        gradient (fc1.weight[:, :D] and fc2.weight only) into G; returns gp (0-dim tensor)."""
        ops = self.ops
        W1f = d['fc1.weight'][:, :D]
        h_hat = ops.gp_interp(h_real, h_fake, alpha.reshape(-1).contiguous())
        v, s = ops.gp_slope_v(h_hat, d['fc2.weight'][0])
        gvec = ops.mm(v, W1f)
        gp = ops.gp_rownorm_(gvec, lam)  # gvec <- d gp / d gvec
        if G is not None:
            dv = ops.mm(gvec, W1f.t())
            ops.mm(v.t(), gvec, out=G['fc1.weight'][:, :D], accumulate=True)
            ops.gp_dw2_(dv, s, G['fc2.weight'][0])
        return gp

    # ------------------------------------------------------------------------------------------ keyword head
    def keyword_fwd_bwd(self, kw, fake, kw_idx, kw_mask, row_weight, G=None):
        """decoder_module.py:39-59.  row_weight [Bg] multiplies each row's loss in the objective (reg_ratio *
        keywords_weight / Bg).  Returns (per-row loss [Bg], d objective / d fake or None); adds head grads into G."""
        ops = self.ops
        xk = ops.mm(fake, kw['proj_fc.weight'].t(), bias=kw['proj_fc.bias'])
        logits = ops.mm(xk, kw['word_emb'].t())
        loss_b = ops.keyword_ce_(logits, kw_idx, kw_mask, row_weight)  # logits <- d objective / d logits
        if G is None:
            return loss_b, None
        dxk = ops.mm(logits, kw['word_emb'])
        ops.mm(dxk.t(), fake, out=G['proj_fc.weight'], accumulate=True)
        ops.colsum(dxk, out=G['proj_fc.bias'])
        dfake = ops.mm(dxk, kw['proj_fc.weight'])
        return loss_b, dfake

    # ------------------------------------------------------------------------------------------ iterations
    def critic_iteration(self, d, g, real, lab, noise, alpha, sim=None, lam=10.0):
        """train_gan.py:228-259 up to optimizer_d.step().  Returns (d_cost, gp, grads for d, fake)."""
        ops = self.ops
        Bg, D = real.shape
        G = zero_grads(d)
        w = torch.full((Bg,), 1.0 / Bg, device=real.device) if sim is None else (sim * (1.0 / Bg)).contiguous()
        lab_half = self.critic_label_half(d, lab, D)
        real_logits, h_real = self.critic_fwd(d, real, lab_half)
        fake, _ = self.generator_fwd(g, noise, lab)
        fake_logits, h_fake = self.critic_fwd(d, fake, lab_half)
        critic_real = ops.weighted_sum(real_logits, w, 1.0)
        critic_fake = ops.weighted_sum(fake_logits, w, 1.0)
        dh_real = self.critic_bwd((-w).contiguous(), h_real, d, G)
        dh_fake = self.critic_bwd(w, h_fake, d, G)
        self.critic_fc1_grads(dh_real, real, None, G)
        self.critic_fc1_grads(dh_fake, fake, None, G)
        dh_sum = dh_real.add_(dh_fake)
        ops.mm(dh_sum.t(), lab, out=G['fc1.weight'][:, D:], accumulate=True)
        ops.colsum(dh_sum, out=G['fc1.bias'])
        gp = self.gradient_penalty(d, h_real, h_fake, alpha, D, G, lam)
        return critic_fake - critic_real, gp, G, fake

    def generator_iteration(self, d, g, kw, lab, noise, sim=None, kw_idx=None, kw_mask=None, reg_ratio=0.0):
        """train_gan.py:262-306 up to optimizer_g.step().  Returns (g_cost, keyword_loss, grads g, grads kw, fake)."""
        ops = self.ops
        Bg = lab.shape[0]
        Gg = zero_grads(g)
        w = torch.full((Bg,), 1.0 / Bg, device=lab.device) if sim is None else (sim * (1.0 / Bg)).contiguous()
        fake, h = self.generator_fwd(g, noise, lab)
        D = fake.shape[1]
        Gk, kw_loss = None, torch.zeros((), device=lab.device)
        dfake = None
        if reg_ratio > 0:
            Gk = zero_grads(kw, skip=('word_emb',))
            kww = torch.ones(Bg, device=lab.device) if sim is None else torch.where(sim != 1, torch.zeros_like(sim), sim)
            loss_b, dfake = self.keyword_fwd_bwd(kw, fake, kw_idx, kw_mask, (kww * (reg_ratio / Bg)).contiguous(), Gk)
            kw_loss = ops.weighted_sum(loss_b, kww.contiguous(), 1.0 / Bg)
        lab_half = self.critic_label_half(d, lab, D)
        fake_logits, h_d = self.critic_fwd(d, fake, lab_half)
        g_cost = -ops.weighted_sum(fake_logits, w, 1.0)
        dh = self.critic_bwd((-w).contiguous(), h_d, d, None)
        W1f = d['fc1.weight'][:, :D]
        if dfake is None:
            dfake = ops.mm(dh, W1f)
        else:
            ops.mm(dh, W1f, out=dfake, accumulate=True)
        self.generator_bwd(dfake, fake, h, g, noise, lab, Gg)
        return g_cost, kw_loss, Gg, Gk, fake


class WganGpStep:
    """One full WGAN-GP step = `critic_iters` critic iterations + one generator iteration with Adam(betas .5/.999),
    as train_gan.py:223-308 orders them.  Parameters live in dicts of tensors (same keys as the reference's
    `generator` / `discriminator` / keyword-head state dicts); torch.optim.Adam does the updates."""

    def __init__(self, ops, g, d, kw=None, lr=1e-4, critic_iters=5, reg_ratio=0.0, lam=10.0, capturable=False):
        self.eng = GanEngine(ops)
        self.g, self.d, self.kw = g, d, kw
        self.critic_iters, self.reg_ratio, self.lam = critic_iters, reg_ratio, lam
        for t in list(g.values()) + list(d.values()):
            t.requires_grad_(False)
        self._gp = {k: torch.nn.Parameter(v, requires_grad=False) for k, v in g.items()}
        self._dp = {k: torch.nn.Parameter(v, requires_grad=False) for k, v in d.items()}
        self._kp = {} if kw is None else {k: torch.nn.Parameter(v, requires_grad=False)
                                          for k, v in kw.items() if k != 'word_emb'}
        # capturable=True keeps Adam's step counters on the device, so a whole step can be recorded into a HIP graph
        # (`graphed_step` below): at the drivers' default batch of 8 the step is ~300 launches of microsecond kernels
        extra = {'capturable': True} if capturable else {}
        if all(t.is_cuda for t in list(g.values()) + list(d.values())):
            extra['fused'] = True  # one multi-tensor launch per step() instead of ~8 elementwise passes (0.6 ms per
            #                        WGAN-GP step at any batch size: 6 optimizer steps x 8 launches of 10-15 us)
        self.opt_d = torch.optim.Adam(list(self._dp.values()), lr=lr, betas=(0.5, 0.999), **extra)
        self.opt_g = torch.optim.Adam(list(self._gp.values()) + list(self._kp.values()), lr=lr, betas=(0.5, 0.999),
                                      **extra)

    def critic_step(self, real, lab, noise, alpha, sim=None):
        d_cost, gp, G, fake = self.eng.critic_iteration(self.d, self.g, real, lab, noise, alpha, sim, self.lam)
        for k, p in self._dp.items():
            p.grad = G[k]
        self.opt_d.step()
        return d_cost, gp, G, fake

    def generator_step(self, lab, noise, sim=None, kw_idx=None, kw_mask=None):
        g_cost, kw_loss, Gg, Gk, fake = self.eng.generator_iteration(self.d, self.g, self.kw, lab, noise, sim, kw_idx,
                                                                     kw_mask, self.reg_ratio)
        for k, p in self._gp.items():
            p.grad = Gg[k]
        for k, p in self._kp.items():
            p.grad = Gk[k]
        self.opt_g.step()
        return g_cost, kw_loss, Gg, Gk, fake


def graphed_step(fn, warmup=3):
    """Record `fn()` (a full WGAN-GP step over STATIC input tensors, built with `capturable=True`) into one HIP graph
    and return a replay callable.  Launch-bound at small batches (train_gan.py's default batch is 8): the eager step is
    a few hundred kernel launches from Python, the replay is one graph launch.  The caller refreshes the static input
    tensors in place between replays; the outputs of `fn` recorded at capture time are returned by every replay."""
    if warmup < 1:  # state created lazily inside the recording (Adam's moments) would be re-zeroed by every replay
        raise ValueError('graphed_step needs at least one eager warm-up step before the recording')
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):  # warm-up off the capture: lazy initialisation (optimizer state, workspaces)
        for _ in range(warmup):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = fn()

    def replay():
        graph.replay()
        return out
    return replay

