"""Shared bodies of the WGAN-GP parity tests (CPU stand-in and GPU), against tests/golden/gan_step_*.npz captured from
the reference in the call order of train_gan.py:223-308.

Tolerances (fp32): costs / penalties rtol 1e-4; gradients rtol 1e-4 (critic: 3e-4, sums over ndh=800 x Bg terms).
The generator iteration is checked TWICE: `g0.*` = at the fixed initial critic, before any optimizer step (plain fp32
tolerance), and `g.*` = after the critic's Adam steps, where the critic parameters may differ from the reference's by
up to lr per step in elements whose gradient is rounding noise (Adam moves an element by ~lr * g / |g|), so quantities
downstream of them get a wider, stated tolerance.
"""
import numpy as np
import torch

from gzsl_text_amd.gan_engine import WganGpStep
from tests.conftest import load_golden
from tests.golden import synth
from tests.helpers import close

GAN_DIMS = {
    'gan_step_tiny': dict(seed=11, n_iters=5, Bg=6, D=16, Ls=16, ngh=24, ndh=20, E=8, Vk=30, top_k=5),
    'gan_step_medium': dict(seed=12, n_iters=2, Bg=96, D=400, Ls=400, ngh=800, ndh=800, E=200, Vk=300, top_k=20),
}
LR = 1e-4


def _cut(a, full):
    a = a.detach().cpu().numpy() if torch.is_tensor(a) else a
    return a if full else a.reshape(a.shape[0], -1)[:4]


def _close_adam(a, b, golden_grads, n_steps, grad_floor, what):
    """Parameters after `n_steps` Adam(lr) steps.  Every element: within the largest possible move difference,
    2 * lr * n_steps.  Elements whose golden gradient exceeded `grad_floor` in EVERY step (the update direction
    m / sqrt(v) is then insensitive to rounding noise in the gradient): tight, 2e-6 + 1e-5 |b|."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    err = np.abs(a - b)
    assert err.max() <= 2.02 * LR * n_steps + 1e-6, f'{what}: max err {err.max():.3e}'
    mask = np.ones(b.shape, bool)
    for gg in golden_grads:
        mask &= np.abs(np.asarray(gg)) > grad_floor
    bad = mask & (err > 2e-6 + 1e-5 * np.abs(b))
    assert not bad.any(), (f'{what}: {int(bad.sum())} of {int(mask.sum())} elements with |golden grad| > {grad_floor} '
                           f'are not tight (max err there {err[mask].max():.3e})')
    return float(mask.mean())


def run_gan_step_check(name, ops, device):
    g = load_golden(name)
    kwargs = dict(GAN_DIMS[name])
    n_iters = kwargs.pop('n_iters')
    s = synth.synth_gan(n_iters=n_iters, **kwargs)
    full = name.endswith('tiny')

    def dev(d):
        return {k: torch.from_numpy(v.copy()).to(device) for k, v in d.items()}

    def t(a):
        return torch.from_numpy(a).to(device)

    step = WganGpStep(ops, dev(s['g']), dev(s['d']), dev(s['kw']), lr=LR, critic_iters=n_iters, reg_ratio=0.01)
    kw_idx, kw_mask = t(s['kw_idx']), t(s['kw_mask'])
    # ---- generator iteration at the FIXED initial critic (no optimizer step before it): plain fp32 tolerance
    I = s['iters'][n_iters]
    g_cost, kw_loss, Gg, Gk, fake = step.eng.generator_iteration(step.d, step.g, step.kw, t(I['lab']), t(I['noise']),
                                                                t(I['sim']), kw_idx, kw_mask, 0.01)
    close(g_cost, g['g0.g_cost'], 1e-4, 2e-6, 'g_cost at the fixed critic')
    close(kw_loss, g['g0.kw_loss'], 1e-5, 2e-5, 'keyword loss at the fixed critic')
    close(fake, g['g0.fake'], 1e-5, 2e-5, 'generated feats at the fixed critic')
    for k in Gg:
        close(_cut(Gg[k], full), g[f'g0.grad.{k}'], 1e-4, 1e-6, f'generator grad {k} at the fixed critic')
    for k in Gk:
        close(_cut(Gk[k], full), g[f'g0.grad.kw.{k}'], 1e-4, 1e-6, f'keyword head grad {k} at the fixed critic')
    # ---- the step proper: n_iters critic iterations (+ Adam) and one generator iteration (+ Adam)
    for it in range(n_iters):
        I = s['iters'][it]
        d_cost, gp, G, fake = step.critic_step(t(I['real']), t(I['lab']), t(I['noise']), t(I['alpha']), t(I['sim']))
        close(d_cost, g[f'c{it}.d_cost'], 1e-4, 2e-5 + 2e-4 * it, 'd_cost')
        close(gp, g[f'c{it}.gp'], 1e-4 + 1e-4 * it, 2e-5, 'gp')
        if f'c{it}.fake' in g:
            close(fake, g[f'c{it}.fake'], 1e-5, 2e-5, 'generated feats')
        for k in G:
            # iteration 0 runs on the initial parameters; later ones on Adam-updated critics (see module docstring)
            close(_cut(G[k], full), g[f'c{it}.grad.{k}'], 3e-4 if it == 0 else 2e-3, 3e-6 if it == 0 else 2e-5,
                  f'critic grad {k} it {it}')
    I = s['iters'][n_iters]
    g_cost, kw_loss, Gg, Gk, fake = step.generator_step(t(I['lab']), t(I['noise']), t(I['sim']), kw_idx, kw_mask)
    close(g_cost, g['g.g_cost'], 1e-4, 1.5e-4 * n_iters, 'g_cost')
    close(kw_loss, g['g.kw_loss'], 1e-5, 2e-5, 'keyword loss')
    close(fake, g['g.fake'], 1e-5, 2e-5, 'generated feats (G step)')
    for k in Gg:
        close(_cut(Gg[k], full), g[f'g.grad.{k}'], 2e-3, 2e-5, f'generator grad {k}')
    for k in Gk:
        close(_cut(Gk[k], full), g[f'g.grad.kw.{k}'], 3e-4, 3e-6, f'keyword head grad {k}')
    cover = []
    for k in step.d:
        cover.append(_close_adam(_cut(step.d[k], full), g[f'after.d.{k}'],
                                 [g[f'c{it}.grad.{k}'] for it in range(n_iters)], n_iters, 1e-5, f'param after step d.{k}'))
    for k in step.g:
        cover.append(_close_adam(_cut(step.g[k], full), g[f'after.g.{k}'], [g[f'g.grad.{k}']], 1, 1e-4,
                                 f'param after step g.{k}'))
    for k in Gk:
        cover.append(_close_adam(_cut(step.kw[k], full), g[f'after.kw.{k}'], [g[f'g.grad.kw.{k}']], 1, 1e-5,
                                 f'param after step kw.{k}'))
    assert np.mean(cover) > 0.3, f'the tight mask covers too little to mean anything: {cover}'


def run_gan_module_api_check(ops, device):
    """The nn.Module surface train_gan.py:243-306 calls (ConditionalGenerator / ConditionalDiscriminator /
    calc_gradient_penalty / LinearKeywordsDecoder with torch autograd around them), in the driver's call order, against
    the golden of the first critic iteration and of the generator iteration at the fixed critic."""
    from gzsl_text_amd.modules.decoder_module import load_decoder
    from gzsl_text_amd.modules.generative_module import (ConditionalDiscriminator, ConditionalGenerator,
                                                         calc_gradient_penalty)
    name = 'gan_step_tiny'
    g = load_golden(name)
    kw = dict(GAN_DIMS[name])
    n_iters = kw.pop('n_iters')
    s = synth.synth_gan(n_iters=n_iters, **kw)
    dd = s['dims']

    def t(a):
        return torch.from_numpy(a.copy()).to(device)

    G = ConditionalGenerator(dd['Ls'], dd['Ls'], dd['ngh'], dd['D'])
    Dn = ConditionalDiscriminator(dd['D'], dd['Ls'], dd['ndh'], 1)
    G.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in s['g'].items()})
    Dn.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in s['d'].items()})
    KW = load_decoder('linear', dd['D'], dd['E'], torch.from_numpy(s['kw']['word_emb'].copy()))
    KW.proj_fc.weight.data.copy_(torch.from_numpy(s['kw']['proj_fc.weight']))
    KW.proj_fc.bias.data.copy_(torch.from_numpy(s['kw']['proj_fc.bias']))
    G, Dn, KW = G.to(device).use_ops(ops), Dn.to(device).use_ops(ops), KW.to(device).use_ops(ops)
    assert set(G.state_dict()) == set(Dn.state_dict()) == {'fc1.weight', 'fc1.bias', 'fc2.weight', 'fc2.bias'}
    one = torch.ones([], device=device)
    real_rand = torch.rand
    I = s['iters'][0]
    real, lab, noise, sim = (t(I[k]) for k in ('real', 'lab', 'noise', 'sim'))
    Dn.zero_grad()
    critic_real = (Dn(real, lab) * sim).mean()
    critic_real.backward(-one)
    fake = torch.relu(G(noise, lab))
    cf = (Dn(fake.detach(), lab) * sim).mean()
    cf.backward(one)
    torch.rand = lambda *a, **k: torch.from_numpy(I['alpha'].copy())   # supply alpha (generative_module.py:24)
    try:
        gp = calc_gradient_penalty(Dn, real, fake.data, lab)
    finally:
        torch.rand = real_rand
    gp.backward()
    close(cf - critic_real, g['c0.d_cost'], 1e-4, 2e-5, 'd_cost')
    close(gp, g['c0.gp'], 1e-4, 2e-5, 'gp')
    close(fake, g['c0.fake'], 1e-5, 2e-5, 'fake')
    for k, p in Dn.named_parameters():
        close(p.grad, g[f'c0.grad.{k}'], 3e-4, 3e-6, 'critic grad ' + k)
    # generator iteration on the un-updated critic (train_gan.py:262-306): golden g0.*
    I = s['iters'][n_iters]
    lab, noise, sim = (t(I[k]) for k in ('lab', 'noise', 'sim'))
    kw_idx, kw_mask = t(s['kw_idx']), t(s['kw_mask'])
    for p in Dn.parameters():
        p.requires_grad = False
    G.zero_grad()
    KW.zero_grad()
    fake = torch.relu(G(noise, lab))
    kl = torch.mean(KW(fake, kw_idx, kw_mask, lab) * sim.masked_fill(sim != 1, 0.))
    g_cost = -(Dn(fake, lab) * sim).mean()
    (g_cost + 0.01 * kl).backward()
    close(g_cost, g['g0.g_cost'], 1e-4, 2e-6, 'g_cost')
    close(kl, g['g0.kw_loss'], 1e-5, 2e-5, 'kw loss')
    close(fake, g['g0.fake'], 1e-5, 2e-5, 'fake (generator iteration)')
    for k, p in G.named_parameters():
        close(p.grad, g[f'g0.grad.{k}'], 1e-4, 1e-6, 'generator grad ' + k)
    for k, p in KW.named_parameters():
        close(p.grad, g[f'g0.grad.kw.{k}'], 1e-4, 1e-6, 'keyword head grad ' + k)


def run_gan_plain_wgan_check(ops, device):
    """The branches the goldens do not reach: no sim weights (train_gan.py without --add_zero) and reg_ratio = 0 (no
    keyword head), one critic + one generator iteration + both Adam steps, against the numpy oracle (oracle/gan.py,
    itself pinned to the reference by tests/test_oracle_golden.py).  fp32 tolerances as in run_gan_step_check."""
    from oracle import gan as og
    s = synth.synth_gan(n_iters=1, **{k: v for k, v in GAN_DIMS['gan_step_tiny'].items() if k != 'n_iters'})
    d = {k: v.copy() for k, v in s['d'].items()}
    gen = {k: v.copy() for k, v in s['g'].items()}
    I, J = s['iters'][0], s['iters'][1]
    d_cost, gp, gd, fake = og.critic_iteration(d, gen, I['real'], I['lab'], I['noise'], I['alpha'], None)
    d_after = {k: og.adam_step(d[k], gd[k].reshape(d[k].shape), {}, LR) for k in d}
    g_cost, kw_loss, gg, kg, gfake = og.generator_iteration(d_after, gen, None, J['lab'], J['noise'], None, None, None, 0.0)
    g_after = {k: og.adam_step(gen[k], gg[k].reshape(gen[k].shape), {}, LR) for k in gen}

    def dev(x):
        return {k: torch.from_numpy(v.copy()).to(device) for k, v in x.items()}

    def t(a):
        return torch.from_numpy(a).to(device)

    step = WganGpStep(ops, dev(s['g']), dev(s['d']), None, lr=LR, critic_iters=1, reg_ratio=0.0)
    dc, gpv, G, fk = step.critic_step(t(I['real']), t(I['lab']), t(I['noise']), t(I['alpha']), None)
    close(dc, d_cost, 1e-4, 2e-5, 'd_cost (no sim)')
    close(gpv, gp, 1e-4, 2e-5, 'gp (no sim)')
    close(fk, fake, 1e-5, 2e-5, 'fake')
    for k in G:
        close(G[k], gd[k].reshape(G[k].shape), 3e-4, 3e-6, f'critic grad {k} (no sim)')
    gc, kl, Gg, Gk, gfk = step.generator_step(t(J['lab']), t(J['noise']))
    assert Gk is None and float(kl) == 0.0
    close(gc, g_cost, 1e-4, 1.5e-4, 'g_cost (no keyword head)')
    for k in Gg:
        close(Gg[k], gg[k].reshape(Gg[k].shape), 2e-3, 2e-5, f'generator grad {k} (no keyword head)')
    for k in step.d:
        _close_adam(_cut(step.d[k], True), d_after[k], [gd[k].reshape(d[k].shape)], 1, 1e-5, f'critic param {k}')
    for k in step.g:
        _close_adam(_cut(step.g[k], True), g_after[k], [gg[k].reshape(gen[k].shape)], 1, 1e-4, f'generator param {k}')


def run_gan_full_step_check(ops, device):
    """WganGpStep.full_step (generator forward of all critic iterations in one stacked pass) lands on the parameters of
    critic_step x n + generator_step: same arithmetic per row, so fp32-tight."""
    name = 'gan_step_tiny'
    kwargs = dict(GAN_DIMS[name])
    n_iters = kwargs.pop('n_iters')
    s = synth.synth_gan(n_iters=n_iters, **kwargs)

    def dev(x):
        return {k: torch.from_numpy(v.copy()).to(device) for k, v in x.items()}

    def t(a):
        return torch.from_numpy(a).to(device)

    its = [{k: t(v) for k, v in I.items()} for I in s['iters']]
    kw_idx, kw_mask = t(s['kw_idx']), t(s['kw_mask'])
    a = WganGpStep(ops, dev(s['g']), dev(s['d']), dev(s['kw']), lr=LR, critic_iters=n_iters, reg_ratio=0.01)
    b = WganGpStep(ops, dev(s['g']), dev(s['d']), dev(s['kw']), lr=LR, critic_iters=n_iters, reg_ratio=0.01)
    b.STACK_MIN_ROWS = 1  # the tiny case has 6 rows: force the stacked generator pass
    want = []
    for i in range(n_iters):
        I = its[i]
        want.append(a.critic_step(I['real'], I['lab'], I['noise'], I['alpha'], I['sim'])[:2])
    I = its[n_iters]
    g_want = a.generator_step(I['lab'], I['noise'], I['sim'], kw_idx, kw_mask)[:2]
    costs, g_got = b.full_step([(I['real'], I['lab'], I['noise'], I['alpha'], I['sim']) for I in its[:n_iters]],
                               (I['lab'], I['noise'], I['sim'], kw_idx, kw_mask))
    for (dc, gp), (dc2, gp2) in zip(want, costs):
        close(dc2, _cut(dc, True), 1e-5, 1e-6, 'd_cost')
        close(gp2, _cut(gp, True), 1e-5, 1e-6, 'gp')
    close(g_got[0], _cut(g_want[0], True), 1e-5, 1e-6, 'g_cost')
    close(g_got[1], _cut(g_want[1], True), 1e-5, 1e-6, 'kw_loss')
    for k in a.d:
        close(b.d[k], _cut(a.d[k], True), 1e-6, 2.02 * LR * n_iters, 'critic parameters ' + k)
        assert float((b.d[k] - a.d[k]).abs().mean()) < 1e-6, k
    for k in a.g:
        close(b.g[k], _cut(a.g[k], True), 1e-6, 2.02 * LR, 'generator parameters ' + k)
        assert float((b.g[k] - a.g[k]).abs().mean()) < 1e-6, k
