"""Module-level GAN API (generative_module / decoder_module / lm_module mirrors) on the CPU stand-in: the call
sequence of train_gan.py:228-307 through the nn.Module surface, against the reference's golden vectors."""
import numpy as np
import torch

from gzsl_text_amd.modules.decoder_module import load_decoder
from gzsl_text_amd.modules.generative_module import (ConditionalDiscriminator, ConditionalGenerator,
                                                     calc_gradient_penalty)
from gzsl_text_amd.modules.lm_module import LabelEncodingCache, RNNLabelEncoder
from tests.conftest import load_golden
from tests.gan_checks import GAN_DIMS
from tests.golden import synth
from tests.helpers import close
from tests.ops_cpu import CpuOps


def test_module_api_follows_train_gan_call_order():
    name = 'gan_step_tiny'
    g = load_golden(name)
    kw = dict(GAN_DIMS[name])
    n_iters = kw.pop('n_iters')
    s = synth.synth_gan(n_iters=n_iters, **kw)
    dd = s['dims']
    ops = CpuOps()
    G = ConditionalGenerator(dd['Ls'], dd['Ls'], dd['ngh'], dd['D']).use_ops(ops)
    Dn = ConditionalDiscriminator(dd['D'], dd['Ls'], dd['ndh'], 1).use_ops(ops)
    G.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in s['g'].items()})
    Dn.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in s['d'].items()})
    KW = load_decoder('linear', dd['D'], dd['E'], torch.from_numpy(s['kw']['word_emb'].copy())).use_ops(ops)
    KW.proj_fc.weight.data.copy_(torch.from_numpy(s['kw']['proj_fc.weight']))
    KW.proj_fc.bias.data.copy_(torch.from_numpy(s['kw']['proj_fc.bias']))
    assert set(G.state_dict()) == set(Dn.state_dict()) == {'fc1.weight', 'fc1.bias', 'fc2.weight', 'fc2.bias'}
    one = torch.ones([])
    real_rand = torch.rand
    I = s['iters'][0]
    real, lab, noise, sim = (torch.from_numpy(I[k]) for k in ('real', 'lab', 'noise', 'sim'))
    Dn.zero_grad()
    (Dn(real, lab) * sim).mean().backward(-one)
    fake = torch.relu(G(noise, lab))
    cf = (Dn(fake.detach(), lab) * sim).mean()
    cf.backward(one)
    torch.rand = lambda *a, **k: torch.from_numpy(I['alpha'].copy())
    try:
        gp = calc_gradient_penalty(Dn, real, fake.data, lab)
    finally:
        torch.rand = real_rand
    gp.backward()
    close(gp, g['c0.gp'], 1e-4, 2e-5, 'gp')
    close(fake, g['c0.fake'], 1e-5, 2e-5, 'fake')
    for k, p in Dn.named_parameters():
        close(p.grad, g[f'c0.grad.{k}'], 3e-4, 3e-6, 'critic grad ' + k)
    # generator step on the un-updated critic is not in the golden; check it against the fused engine instead
    from gzsl_text_amd.gan_engine import GanEngine
    eng = GanEngine(ops)
    dpar = {k: v.detach() for k, v in Dn.named_parameters()}
    gpar = {k: v.detach() for k, v in G.named_parameters()}
    kpar = {'proj_fc.weight': KW.proj_fc.weight.detach(), 'proj_fc.bias': KW.proj_fc.bias.detach(), 'word_emb': KW.word_emb}
    kw_idx, kw_mask = torch.from_numpy(s['kw_idx']), torch.from_numpy(s['kw_mask'])
    g_cost_e, kwl_e, Gg, Gk, _ = eng.generator_iteration(dpar, gpar, kpar, lab, noise, sim, kw_idx, kw_mask, 0.01)
    for p in Dn.parameters():
        p.requires_grad = False
    G.zero_grad()
    KW.zero_grad()
    fake = torch.relu(G(noise, lab))
    kl = torch.mean(KW(fake, kw_idx, kw_mask, lab) * sim.masked_fill(sim != 1, 0.))
    g_cost = -(Dn(fake, lab) * sim).mean()
    (g_cost + 0.01 * kl).backward()
    close(g_cost, g_cost_e, 1e-5, 1e-6, 'g_cost')
    close(kl, kwl_e, 1e-5, 1e-6, 'kw loss')
    for k, p in G.named_parameters():
        close(p.grad, Gg[k], 1e-4, 1e-6, 'generator grad ' + k)
    for k, p in KW.named_parameters():
        close(p.grad, Gk[k], 1e-4, 1e-6, 'keyword head grad ' + k)


def test_label_rnn_mirror_and_cache_match_reference():
    g = load_golden('label_rnn_tiny')
    enc = RNNLabelEncoder(g['word_emb'].copy())
    for k, p in enc.rnn.named_parameters():
        p.data.copy_(torch.from_numpy(g['rnn.' + k]))
    assert {'encoder.weight', 'decoder.weight', 'rnn.weight_ih_l0', 'rnn.weight_hh_l0', 'rnn.bias_ih_l0',
            'rnn.bias_hh_l0'} == set(enc.state_dict())
    ids, m = torch.from_numpy(g['ids']), torch.from_numpy(g['mask'])
    with torch.no_grad():
        for mode in ('max', 'mean'):
            close(enc.forward_enc(ids, m, None, training=False, pool_mode=mode), g['pool_' + mode], 1e-5, 1e-6, mode)
    cache = LabelEncodingCache(enc, ids, m, trunc=20, pool_mode='max', chunk=4)
    sel = torch.tensor([3, 0, 3, 10])
    close(cache(sel), g['pool_max'][sel.numpy()], 1e-5, 1e-6, 'cached label encoding')
    # the packed-GRU orchestration the HIP path uses (gather, one input GEMM, per-step GEMM + masked gate/pool step)
    from gzsl_text_amd.modules.lm_module import forward_enc_hip
    from tests.ops_cpu import CpuOps
    for mode in ('max', 'mean'):
        close(forward_enc_hip(enc, ids, m, mode, CpuOps()), g['pool_' + mode], 1e-5, 1e-6, 'packed GRU steps ' + mode)
    cache2 = LabelEncodingCache(enc, ids, m, trunc=20, pool_mode='max', chunk=4, ops=CpuOps())
    close(cache2(sel), g['pool_max'][sel.numpy()], 1e-5, 1e-6, 'cache filled through the step kernels')
