"""Shared body of the WGAN-GP step parity test (CPU stand-in and GPU): one step in the reference's call order against
tests/golden/gan_step_*.npz."""
import numpy as np
import torch

from gzsl_text_amd.gan_engine import WganGpStep
from tests.conftest import load_golden
from tests.golden import synth
from tests.helpers import close

GAN_DIMS = {
    'gan_step_tiny': dict(seed=11, n_iters=2, Bg=6, D=16, Ls=16, ngh=24, ndh=20, E=8, Vk=30, top_k=5),
    'gan_step_medium': dict(seed=12, n_iters=2, Bg=96, D=400, Ls=400, ngh=800, ndh=800, E=200, Vk=300, top_k=20),
}


def _cut(a, full):
    a = a.detach().cpu().numpy() if torch.is_tensor(a) else a
    return a if full else a.reshape(a.shape[0], -1)[:4]


def _close_adam(a, b, max_move, what):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    err = np.abs(a - b)
    assert err.max() <= 2.02 * max_move + 1e-6, f'{what}: max err {err.max():.3e}'
    tight = err <= 2e-6 + 1e-5 * np.abs(b)
    # (the critic's fc2.bias is ONE element whose gradient is sum(-w) + sum(+w) = rounding noise: bound check only)
    assert a.size < 8 or tight.mean() >= 0.80, f'{what}: only {tight.mean():.3f} tight'


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

    step = WganGpStep(ops, dev(s['g']), dev(s['d']), dev(s['kw']), lr=1e-4, critic_iters=n_iters, reg_ratio=0.01)
    for it in range(n_iters):
        I = s['iters'][it]
        d_cost, gp, G, fake = step.critic_step(t(I['real']), t(I['lab']), t(I['noise']), t(I['alpha']), t(I['sim']))
        close(d_cost, g[f'c{it}.d_cost'], 1e-4, 2e-5, 'd_cost')
        close(gp, g[f'c{it}.gp'], 1e-4, 2e-5, 'gp')
        if f'c{it}.fake' in g:
            close(fake, g[f'c{it}.fake'], 1e-5, 2e-5, 'generated feats')
        for k in G:
            close(_cut(G[k], full), g[f'c{it}.grad.{k}'], 3e-4, 3e-6, f'critic grad {k} it {it}')
    I = s['iters'][n_iters]
    g_cost, kw_loss, Gg, Gk, fake = step.generator_step(t(I['lab']), t(I['noise']), t(I['sim']), t(s['kw_idx']),
                                                        t(s['kw_mask']))
    # The generator iteration runs on critic parameters that went through Adam steps: Adam moves every element by
    # ~lr * sign(grad), so critic entries whose gradient is rounding noise may sit lr = 1e-4 away from the
    # reference's (see _close_adam).  Quantities downstream of the critic get a correspondingly wider tolerance.
    close(g_cost, g['g.g_cost'], 1e-4, 3e-4, 'g_cost')
    close(kw_loss, g['g.kw_loss'], 1e-5, 2e-5, 'keyword loss')
    close(fake, g['g.fake'], 1e-5, 2e-5, 'generated feats (G step)')
    for k in Gg:
        close(_cut(Gg[k], full), g[f'g.grad.{k}'], 2e-3, 2e-5, f'generator grad {k}')
    for k in Gk:
        close(_cut(Gk[k], full), g[f'g.grad.kw.{k}'], 3e-4, 3e-6, f'keyword head grad {k}')
    for tag, prm, n in (('d', step.d, n_iters), ('g', step.g, 1)):
        for k in prm:
            _close_adam(_cut(prm[k], full), g[f'after.{tag}.{k}'], 1e-4 * n, f'param after step {tag}.{k}')
    for k in Gk:
        _close_adam(_cut(step.kw[k], full), g[f'after.kw.{k}'], 1e-4, f'param after step kw.{k}')
