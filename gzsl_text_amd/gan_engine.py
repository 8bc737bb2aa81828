"""WGAN-GP feature generator / critic step on HIP kernels (the hot loop of train_gan.py:223-308).

Parameters are dicts keyed like the reference's state dicts:
  generator / discriminator: fc1.weight [h, a+b], fc1.bias, fc2.weight [o, h], fc2.bias   (generative_module.py:41-68)
  keyword head: proj_fc.weight [E, D], proj_fc.bias, word_emb [Vk, E] (frozen buffer)       (decoder_module.py:25-31)

What differs from "run the reference's ops one by one":
  * The label half of the critic's fc1 (lab @ W1[:, D:]^T + b1) is shared by D(real), D(fake) and the penalty.
  * The gradient penalty uses the closed form (no double backward), and its fc1 pre-activation is the alpha-blend of
    the real / fake pre-activations (fc1 is affine and the label half is common), so it costs no extra fc1 GEMM.
  * Building blocks used by the nn.Module mirrors (modules/generative_module.py, decoder_module.py): fc1 as two GEMMs
    over the column blocks of fc1.weight (no torch.cat), the critic's single-output fc2 as a fused row-dot with
    LeakyReLU applied on the fly.
  * Whole iterations (`critic_iteration`, `generator_iteration`, what `WganGpStep` runs): rows are stacked so that every
    product is one GEMM, everything between the GEMMs is one kernel, parameters / gradients / Adam moments live in flat
    buffers with one layout (see the comment above `critic_iteration`).
Sign conventions follow the driver: critic grads = d(-critic_real + critic_fake + gp), generator grads =
d(-critic_fake + reg_ratio * keyword_loss) (train_gan.py:244-256, 300-306).
"""
import math

import torch

from .hip import ACT_LRELU, ACT_NONE, ACT_RELU


class Layout:
    """Where each tensor of an ordered (key, shape) list sits in ONE flat fp32 buffer.  Every offset is a multiple of 64
    floats (256 bytes), so views are valid operands for the 16-byte vector paths of the kernels.  Parameters, their
    gradients and Adam's moments share a layout: one fill zeroes all gradients of an iteration, one launch updates all
    parameters (gzsl_adam_flat_step)."""
    ALIGN = 64

    def __init__(self, items):
        self.offs, off = {}, 0
        for k, shape in items:
            n = math.prod(shape)
            self.offs[k] = (off, n, tuple(shape))
            off += -(-n // self.ALIGN) * self.ALIGN
        self.size = off

    @classmethod
    def of(cls, params, skip=(), prefix=''):
        return cls([(prefix + k, v.shape) for k, v in params.items() if k not in skip])

    def views(self, flat):
        return {k: flat[o:o + n].view(shape) for k, (o, n, shape) in self.offs.items()}

    def __add__(self, other):
        out = Layout([])
        out.offs = dict(self.offs)
        for k, (o, n, shape) in other.offs.items():
            out.offs[k] = (self.size + o, n, shape)
        out.size = self.size + other.size
        return out


N_SCALARS = 64  # tail of every gradient buffer: [0], [1] hold the iteration's two scalar results


def zero_grads(layout, device):
    """(flat gradient buffer incl. the scalar tail, views by key, scalar tail) -- one fill launch for all of it."""
    flat = torch.zeros(layout.size + N_SCALARS, dtype=torch.float32, device=device)
    return flat[:layout.size], layout.views(flat), flat[layout.size:]


class _Transposed:
    """W^T as the right-hand operand of `x @ W^T` for nn.Linear weights W [N, K].  The strided view W.t() makes the GEMM
    read B along k (64-byte pieces of 208 weight rows per k-tile); from 8192 rows of x on, a materialised [K, N] copy
    (B contiguous along its rows) is 15-28 % faster at the step's widths (tools/bench_gan_gemm.py: 16384 x 800 x 400 in
    107 instead of 146 us) and the 6 us copy is paid back by the first product.  One copy per weight and iteration."""
    RC_ROWS = 8192

    def __init__(self):
        self._t = {}

    def __call__(self, W, rows, k0=0, k1=None):
        """(W[:, k0:k1])^T for a product with `rows` rows on the left."""
        if rows < self.RC_ROWS:
            return W[:, k0:k1].t()
        if id(W) not in self._t:
            self._t[id(W)] = W.t().contiguous()
        return self._t[id(W)][k0:k1]


class GanEngine:
    def __init__(self, ops):
        self.ops = ops

    # ------------------------------------------------------------------------------------------ generator
    def generator_fwd(self, g, noise, lab, relu_out=True, out=None, tr=None):
        """relu(G(noise, lab)) (generative_module.py:48-53 + train_gan.py:248).  Returns (fake, hidden activations)."""
        ops = self.ops
        tr = tr or _Transposed()
        Bg, n1 = noise.shape
        W1 = g['fc1.weight']
        h = ops.mm(noise, tr(W1, Bg, 0, n1), bias=g['fc1.bias'])
        ops.mm(lab, tr(W1, Bg, n1), out=h, accumulate=True, act=ACT_LRELU)
        fake = ops.mm(h, tr(g['fc2.weight'], Bg), out=out, bias=g['fc2.bias'], act=ACT_RELU if relu_out else ACT_NONE)
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

    def critic_fwd(self, d, feat, lab_half, consume=False, need_logits=True):
        """Returns (logits [Bg] or None, fc1 pre-activations [Bg, ndh]).  consume=True: `lab_half` is not needed again and
        becomes the pre-activation buffer (no copy)."""
        ops = self.ops
        D = feat.shape[1]
        h = lab_half if consume else lab_half.clone()
        ops.mm(feat, d['fc1.weight'][:, :D].t(), out=h, accumulate=True)
        logits = ops.rowdot(h, d['fc2.weight'][0], d['fc2.bias'], lrelu_in=True) if need_logits else None
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
    # The whole-iteration forms below are what WganGpStep runs.  They stack rows so that each product is ONE GEMM:
    #   F   = [real; fake; d gp/d gvec]         [3Bg, D]     hf = F[:2Bg] @ W1f^T  (both critic evaluations)
    #   lst = [dh_real; dh_fake; v]             [3Bg, ndh]   d fc1.weight[:, :D] = lst^T @ F  (K = 3Bg, no accumulation)
    # and run everything between those GEMMs in one kernel (gzsl_gan_critic_head / gzsl_gan_gen_head).  Every gradient
    # tensor is written by exactly one GEMM / column sum, or accumulated by kernels ordered on the stream.
    def critic_iteration(self, d, g, real, lab, noise, alpha, sim=None, lam=10.0, fake=None):
        """train_gan.py:228-259 up to optimizer_d.step().  Returns (d_cost, gp, grads for d, fake); the grads dict
        carries `.flat`, the buffer its tensors are views of (laid out as Layout.of(d)).  fake: relu(G(noise, lab)) when
        the caller has it already (WganGpStep.full_step runs the generator once for all critic iterations)."""
        ops = self.ops
        Bg, D = real.shape
        flat, views, scal = zero_grads(Layout.of(d), real.device)
        G = GradDict(views, flat)
        W1, w2 = d['fc1.weight'], d['fc2.weight'][0]
        W1f, tr = W1[:, :D], _Transposed()
        F = torch.empty((3 * Bg, D), dtype=torch.float32, device=real.device)
        F[:Bg].copy_(real)
        if fake is None:
            fake = F[Bg:2 * Bg]
            self.generator_fwd(g, noise, lab, out=fake, tr=tr)
        else:
            F[Bg:2 * Bg].copy_(fake)
            fake = F[Bg:2 * Bg]
        lh = ops.mm(lab, tr(W1, Bg, D), bias=d['fc1.bias'])
        hf = ops.mm(F[:2 * Bg], tr(W1, 2 * Bg, 0, D))
        lst, dhs, sl = ops.gan_critic_head(hf, lh, w2, sim, alpha.reshape(-1), G['fc2.weight'][0], G['fc1.bias'],
                                           scal[0:1])
        gvec = ops.mm(lst[2 * Bg:], W1f, out=F[2 * Bg:])
        ops.gp_rownorm_(gvec, lam, out=scal[1:2])  # gvec <- d gp / d gvec
        dv = ops.mm(gvec, tr(W1, Bg, 0, D))
        ops.gp_dw2_(dv, sl, G['fc2.weight'][0])
        ops.mm(lst.t(), F, out=G['fc1.weight'][:, :D])
        ops.mm(dhs.t(), lab, out=G['fc1.weight'][:, D:])
        return scal[0], scal[1], G, fake

    def generator_iteration(self, d, g, kw, lab, noise, sim=None, kw_idx=None, kw_mask=None, reg_ratio=0.0):
        """train_gan.py:262-306 up to optimizer_g.step().  Returns (g_cost, keyword_loss, grads g, grads kw, fake); the
        two grads dicts are views of ONE flat buffer (`Gg.flat`, laid out as Layout.of(g) + Layout.of(kw head))."""
        ops = self.ops
        Bg = lab.shape[0]
        use_kw = reg_ratio > 0
        lay = Layout.of(g, prefix='g.')
        if use_kw:
            lay = lay + Layout.of(kw, skip=('word_emb',), prefix='kw.')
        flat, views, scal = zero_grads(lay, lab.device)
        Gg = GradDict({k[2:]: v for k, v in views.items() if k.startswith('g.')}, flat)
        Gk = GradDict({k[3:]: v for k, v in views.items() if k.startswith('kw.')}, flat) if use_kw else None
        tr = _Transposed()
        xin = torch.cat([noise, lab], 1)
        h = ops.mm(xin, tr(g['fc1.weight'], Bg), bias=g['fc1.bias'], act=ACT_LRELU)
        fake = ops.mm(h, tr(g['fc2.weight'], Bg), bias=g['fc2.bias'], act=ACT_RELU)
        D = fake.shape[1]
        dfake = None
        if use_kw:  # decoder_module.py:39-59 with train_gan.py:288-296's row weights
            Wp, We = kw['proj_fc.weight'], kw['word_emb']
            xk = ops.mm(fake, tr(Wp, Bg), bias=kw['proj_fc.bias'])
            logits = ops.mm(xk, We.t())
            ops.keyword_ce_sim_(logits, kw_idx, kw_mask, sim, reg_ratio / Bg, 1.0 / Bg, scal[1:2])  # logits <- d / d logits
            dxk = ops.mm(logits, We)
            ops.mm(dxk.t(), fake, out=Gk['proj_fc.weight'])
            ops.colsum(dxk, out=Gk['proj_fc.bias'])
            dfake = ops.mm(dxk, Wp)
        W1f = d['fc1.weight'][:, :D]
        lh = ops.mm(lab, tr(d['fc1.weight'], Bg, D), bias=d['fc1.bias'])
        hf = ops.mm(fake, tr(d['fc1.weight'], Bg, 0, D))
        dh = ops.gan_gen_head(hf, lh, d['fc2.weight'][0], d['fc2.bias'], sim, scal[0:1])
        dfake = ops.mm(dh, W1f, out=dfake, accumulate=dfake is not None)
        ops.act_bwd_(dfake, fake, ACT_RELU)
        ops.mm(dfake.t(), h, out=Gg['fc2.weight'])
        ops.colsum(dfake, out=Gg['fc2.bias'])
        dhg = ops.mm(dfake, g['fc2.weight'])
        ops.act_bwd_(dhg, h, ACT_LRELU)
        ops.mm(dhg.t(), xin, out=Gg['fc1.weight'])
        ops.colsum(dhg, out=Gg['fc1.bias'])
        return scal[0], scal[1], Gg, Gk, fake


class GradDict(dict):
    """Gradient tensors by parameter name + `.flat`, the one buffer they are views of."""

    def __init__(self, views, flat):
        super().__init__(views)
        self.flat = flat


class WganGpStep:
    """One full WGAN-GP step = `critic_iters` critic iterations + one generator iteration with Adam(betas .5/.999),
    as train_gan.py:223-308 orders them (optimizers: train_gan.py:155-157).

    The parameter dicts passed in (same keys as the reference's `generator` / `discriminator` / keyword-head state
    dicts) are MOVED into flat buffers: read the live parameters from `.g`, `.d`, `.kw` afterwards.  The critic is one
    buffer, generator + keyword head another (the reference gives those two to one optimizer), so an optimizer step is
    one launch of gzsl_adam_flat_step; its step counter lives on the device, which makes the whole step recordable into
    a HIP graph (`graphed_step` below) without further switches."""

    def __init__(self, ops, g, d, kw=None, lr=1e-4, critic_iters=5, reg_ratio=0.0, lam=10.0, betas=(0.5, 0.999),
                 eps=1e-8, capturable=None):
        self.eng, self.ops = GanEngine(ops), ops
        self.critic_iters, self.reg_ratio, self.lam = critic_iters, reg_ratio, lam
        self.lr, self.betas, self.eps = lr, betas, eps
        dev = d['fc1.weight'].device
        self._use_kw = kw is not None and reg_ratio > 0
        self.d, self._pd = self._rehome(Layout.of(d), {k: v for k, v in d.items()}, dev)
        lay_g, src = Layout.of(g, prefix='g.'), {'g.' + k: v for k, v in g.items()}
        if self._use_kw:
            lay_g = lay_g + Layout.of(kw, skip=('word_emb',), prefix='kw.')
            src.update({'kw.' + k: v for k, v in kw.items() if k != 'word_emb'})
        views, self._pg = self._rehome(lay_g, src, dev)
        self.g = {k[2:]: v for k, v in views.items() if k.startswith('g.')}
        self.kw = kw
        if self._use_kw:
            self.kw = {k[3:]: v for k, v in views.items() if k.startswith('kw.')}
            self.kw['word_emb'] = kw['word_emb']
        self._opt = {}
        for name, p in (('d', self._pd), ('g', self._pg)):
            self._opt[name] = (torch.zeros_like(p), torch.zeros_like(p),
                               torch.zeros(ops.ADAM_STATE_INTS, dtype=torch.int32, device=dev))

    @staticmethod
    def _rehome(layout, src, dev):
        flat = torch.zeros(layout.size, dtype=torch.float32, device=dev)
        views = layout.views(flat)
        for k, v in views.items():
            v.copy_(src[k])
        return views, flat

    def _adam(self, name, p, gflat):
        m, v, state = self._opt[name]
        self.ops.adam_flat_step_(p, gflat, m, v, state, self.lr, self.betas[0], self.betas[1], self.eps)

    def critic_step(self, real, lab, noise, alpha, sim=None):
        d_cost, gp, G, fake = self.eng.critic_iteration(self.d, self.g, real, lab, noise, alpha, sim, self.lam)
        self._adam('d', self._pd, G.flat)
        return d_cost, gp, G, fake

    def full_step(self, critic_batches, gen_batch):
        """One WGAN-GP step with all of its batches in hand: critic_batches = list of (real, lab, noise, alpha, sim),
        gen_batch = (lab, noise, sim, kw_idx, kw_mask).  The generator does not change while the critic trains
        (train_gan.py:225-259 only steps optimizer_d), so its forward for ALL critic iterations is one stacked pass
        (three GEMMs over critic_iters * Bg rows instead of 3 * critic_iters over Bg rows).  Same results as
        critic_step x n + generator_step.  Returns (list of (d_cost, gp), (g_cost, kw_loss))."""
        Bg = critic_batches[0][0].shape[0]
        # stacking pays where a Bg-row product leaves the GPU underfilled but is past the skinny kernel's 32 rows
        # (measured: Bg=4096 3.47 -> 3.35 ms; Bg=8 would trade five 5-us skinny products for a 25-us tiled one)
        stack = self.STACK_MIN_ROWS <= Bg < self.STACK_MAX_ROWS
        if stack:
            noise_all = torch.cat([b[2] for b in critic_batches], 0)
            lab_all = torch.cat([b[1] for b in critic_batches], 0)
            fake_all, _ = self.eng.generator_fwd(self.g, noise_all, lab_all)
        costs = []
        for i, (real, lab, noise, alpha, sim) in enumerate(critic_batches):
            d_cost, gp, G, _ = self.eng.critic_iteration(self.d, self.g, real, lab, noise, alpha, sim, self.lam,
                                                         fake=fake_all[i * Bg:(i + 1) * Bg] if stack else None)
            self._adam('d', self._pd, G.flat)
            costs.append((d_cost, gp))
        g_cost, kw_loss = self.generator_step(*gen_batch)[:2]
        return costs, (g_cost, kw_loss)

    STACK_MIN_ROWS, STACK_MAX_ROWS = 64, 8192

    def generator_step(self, lab, noise, sim=None, kw_idx=None, kw_mask=None):
        g_cost, kw_loss, Gg, Gk, fake = self.eng.generator_iteration(self.d, self.g, self.kw, lab, noise, sim, kw_idx,
                                                                     kw_mask, self.reg_ratio if self._use_kw else 0.0)
        self._adam('g', self._pg, Gg.flat)
        return g_cost, kw_loss, Gg, Gk, fake


def graphed_step(fn, warmup=3):
    """Record `fn()` (a full WGAN-GP step of a WganGpStep over STATIC input tensors) into one HIP graph
    and return a replay callable.  Launch-bound at small batches (train_gan.py's default batch is 8): the eager step is
    a few hundred kernel launches from Python, the replay is one graph launch.  The caller refreshes the static input
    tensors in place between replays; the outputs of `fn` recorded at capture time are returned by every replay."""
    if warmup < 1:  # lazily created state (library handles, workspaces) must exist before the recording
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

