"""Pins the CPU oracle (oracle/*.py) to the golden vectors captured from the reference.

Tolerances (fp32): logits/loss rtol 1e-5 atol 1e-5; gradients rtol 1e-4 atol 1e-6
(relaxed to atol 2e-6 on the medium case where sums run over ~600 terms).
"""
import numpy as np
import pytest

from oracle import base_model as om
from oracle import gan as og
from tests.conftest import load_golden
from tests.golden import synth


def _close(a, b, rtol, atol, what):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = np.abs(a - b)
    bad = err > atol + rtol * np.abs(b)
    assert not bad.any(), f'{what}: max err {err.max():.3e} at ref {np.abs(b).max():.3e}, {bad.sum()} bad'


@pytest.mark.parametrize('name', list(synth.BASE_CASES))
@pytest.mark.parametrize('train', [False, True])
def test_base_model_matches_reference(name, train):
    g = load_golden(f"{name}_{'train' if train else 'eval'}")
    case = synth.synth_base(**synth.BASE_CASES[name])
    d = case['dims']
    p = case['params']
    assert np.isclose(synth.checksum(*[p[k] for k in sorted(p)], case['x'], case['y'], case['mask']),
                      g['input_checksum'], rtol=1e-12), 'regenerated inputs drifted from the captured ones'
    margin = om.class_margin_from_count(case['target_count'])
    _close(margin, g['class_margin'], 1e-6, 0, 'class_margin')
    out = om.base_forward_backward(p, case['bufs'], case['x'], case['y'], case['mask'], d['L'], margin, 2.0,
                                   graph_encoder=case['encoder'], drop_seed=synth.DROP_SEED if train else None)
    _close(out['avg'], g['avg'], 1e-5, 1e-6, 'avg label emb')
    _close(out['graph'], g['graph'], 1e-5, 1e-6, 'graph label emb')
    _close(out['logits'], g['logits'], 1e-5, 1e-5, 'logits')
    _close(out['loss'], g['loss'], 1e-5, 1e-5, 'loss')
    if not train:
        _close(out['direct'], g['direct'], 1e-5, 1e-5, 'direct feats')
        _close(out['attn'], g['attn'], 1e-5, 1e-5, 'attn feats')
    atol = 2e-6 if d['T'] > 64 else 1e-6
    for k in p:
        _close(out['grads'][k], g['grad.' + k], 1e-4, atol, 'grad ' + k)


def test_label_chunking_is_exact():
    case = synth.synth_base(**synth.BASE_CASES['base_gate_tiny'])
    d = case['dims']
    margin = om.class_margin_from_count(case['target_count'])
    a = om.base_forward_backward(case['params'], case['bufs'], case['x'], case['y'], case['mask'], d['L'],
                                 margin, 2.0, drop_seed=7)
    b = om.base_forward_backward(case['params'], case['bufs'], case['x'], case['y'], case['mask'], d['L'],
                                 margin, 2.0, drop_seed=7, l_chunk=5)
    _close(a['logits'], b['logits'], 1e-6, 1e-6, 'logits')
    for k in a['grads']:
        _close(a['grads'][k], b['grads'][k], 1e-5, 1e-6, k)


def test_gan_model_variant_matches_reference():
    g = load_golden('gan_model_tiny')
    case = synth.synth_base(**synth.BASE_CASES['base_gate_tiny'])
    d = case['dims']
    p = dict(case['params'])
    margin = om.class_margin_from_count(case['target_count'])
    avg, graph, _ = om.label_embedding_fwd(p, case['bufs'])
    _close(avg, g['frozen_avg'], 1e-5, 1e-6, 'frozen avg')
    _close(graph, g['frozen_graph'], 1e-5, 1e-6, 'frozen graph')
    p['output_fc.weight'] = g['fc0']
    full = om.gan_model_forward_backward(p, avg, case['x'], case['y'], case['mask'], d['L'], margin, 2.0)
    _close(full['logits'], g['full_logits'], 1e-5, 1e-5, 'full logits')
    _close(full['loss'], g['full_loss'], 1e-5, 1e-5, 'full loss')
    _close(full['dfc'], g['full_dfc'], 1e-4, 1e-6, 'full dfc')
    _close(full['direct'], g['real_direct'], 1e-5, 1e-5, 'forward_real_feats direct')
    _close(full['attn'], g['real_attn'], 1e-5, 1e-5, 'forward_real_feats attn')
    sub = om.gan_model_forward_backward(p, avg, case['x'], case['y'], case['mask'], d['L'], margin, 2.0,
                                        label_indices=g['sub'])
    _close(sub['logits'], g['sub_logits'], 1e-5, 1e-5, 'subset logits')
    _close(sub['loss'], g['sub_loss'], 1e-5, 1e-5, 'subset loss')
    _close(sub['dfc'], g['sub_dfc'], 1e-4, 1e-6, 'subset dfc')
    # forward_final + AdamW (two steps)
    fc = g['fc0'].copy()
    m = np.zeros_like(fc)
    v = np.zeros_like(fc)
    Bs = g['ff_feats'].shape[0]
    for step in (1, 2):
        loss, dfc, _ = om.forward_final(fc, g['ff_feats'], g['ff_labels'], margin, 2.0)
        _close(loss / Bs, g[f'ff_loss{step}'], 1e-5, 1e-5, f'forward_final loss {step}')
        _close(dfc / Bs, g[f'ff_dfc{step}'], 1e-4, 1e-6, f'forward_final grad {step}')
        fc, m, v = om.adamw_nobias_step(fc, dfc / np.float32(Bs), m, v, 1e-3, 0.5, 0.999, 1e-6, 5e-4)
        _close(fc, g[f'ff_fc_after{step}'], 1e-5, 1e-6, f'AdamW params after {step}')


@pytest.mark.parametrize('name', list(synth.GAN_TRAIN_CASES))
def test_gan_model_variant_train_mode_matches_reference(name):
    """finetune.py:258,271: ConvLabelAttnGAN.forward in train mode (both dropouts), full list then a label subset."""
    g = load_golden('gan_model_train_' + name)
    case, fc0, sub = synth.gan_train_inputs(name)
    assert np.isclose(synth.checksum(*[case['params'][k] for k in sorted(case['params'])], case['x'], case['y'],
                                     case['mask'], fc0, sub), g['input_checksum'], rtol=1e-12)
    d = case['dims']
    p = dict(case['params'])
    p['output_fc.weight'] = fc0
    margin = om.class_margin_from_count(case['target_count'])
    avg, _, _ = om.label_embedding_fwd(p, case['bufs'])
    full = om.gan_model_forward_backward(p, avg, case['x'], case['y'], case['mask'], d['L'], margin, 2.0,
                                         drop_seed=synth.DROP_SEED)
    _close(full['logits'], g['full_logits'], 1e-5, 2e-5, 'train full logits')
    _close(full['loss'], g['full_loss'], 1e-5, 1e-5, 'train full loss')
    _close(full['dfc'], g['full_dfc'], 1e-4, 1e-6, 'train full dfc')
    part = om.gan_model_forward_backward(p, avg, case['x'], case['y'], case['mask'], d['L'], margin, 2.0,
                                         label_indices=sub, drop_seed=synth.DROP_SEED + 1)
    _close(part['logits'], g['sub_logits'], 1e-5, 2e-5, 'train subset logits')
    _close(part['loss'], g['sub_loss'], 1e-5, 1e-5, 'train subset loss')
    _close(part['dfc'][sub], g['sub_dfc_rows'], 1e-4, 1e-6, 'train subset dfc rows')
    assert not np.delete(part['dfc'], sub, axis=0).any()


from tests.gan_checks import GAN_DIMS, _close_adam  # noqa: E402  (Adam-aware parameter comparison, explicit mask)


def _cut(a, full):
    return a if full else a.reshape(a.shape[0], -1)[:4]


@pytest.mark.parametrize('name', list(GAN_DIMS))
def test_wgan_gp_step_matches_reference(name):
    g = load_golden(name)
    kw = dict(GAN_DIMS[name])
    n_iters = kw.pop('n_iters')
    s = synth.synth_gan(n_iters=n_iters, **kw)
    full = name.endswith('tiny')
    assert np.isclose(synth.checksum(*[s['g'][k] for k in sorted(s['g'])], *[s['d'][k] for k in sorted(s['d'])],
                                     s['iters'][0]['real'], s['iters'][-1]['noise']), g['input_checksum'], rtol=1e-12)
    d = {k: v.copy() for k, v in s['d'].items()}
    gen = {k: v.copy() for k, v in s['g'].items()}
    kwp = {k: v.copy() for k, v in s['kw'].items()}
    dstate = {k: {} for k in d}
    # generator iteration at the fixed initial critic (golden g0.*): plain fp32 tolerance
    I = s['iters'][n_iters]
    g_cost, kw_loss, gg, kg, fake = og.generator_iteration(d, gen, kwp, I['lab'], I['noise'], I['sim'],
                                                           s['kw_idx'], s['kw_mask'], 0.01)
    _close(g_cost, g['g0.g_cost'], 1e-4, 2e-6, 'g_cost at the fixed critic')
    _close(kw_loss, g['g0.kw_loss'], 1e-5, 1e-5, 'keyword loss at the fixed critic')
    for k in gen:
        _close(_cut(gg[k], full), g[f'g0.grad.{k}'], 1e-4, 1e-6, f'generator grad {k} at the fixed critic')
    for k in kg:
        _close(_cut(kg[k], full), g[f'g0.grad.kw.{k}'], 1e-4, 1e-6, f'keyword head grad {k} at the fixed critic')
    for it in range(n_iters):
        I = s['iters'][it]
        d_cost, gp, grads, fake = og.critic_iteration(d, gen, I['real'], I['lab'], I['noise'], I['alpha'], I['sim'])
        _close(d_cost, g[f'c{it}.d_cost'], 1e-4, 1e-5 + 2e-4 * it, 'd_cost')
        _close(gp, g[f'c{it}.gp'], 1e-4 + 1e-4 * it, 1e-5, 'gp')
        if f'c{it}.fake' in g:
            _close(fake, g[f'c{it}.fake'], 1e-5, 1e-5, 'generated feats')
        for k in d:
            _close(_cut(grads[k], full), g[f'c{it}.grad.{k}'], 2e-4 if it == 0 else 2e-3, 2e-6 if it == 0 else 2e-5,
                   f'critic grad {k} it {it}')
            d[k] = og.adam_step(d[k], grads[k], dstate[k], 1e-4)
    I = s['iters'][n_iters]
    g_cost, kw_loss, gg, kg, fake = og.generator_iteration(d, gen, kwp, I['lab'], I['noise'], I['sim'],
                                                           s['kw_idx'], s['kw_mask'], 0.01)
    _close(g_cost, g['g.g_cost'], 1e-4, 1.5e-4 * n_iters, 'g_cost')
    _close(kw_loss, g['g.kw_loss'], 1e-5, 1e-5, 'keyword loss')
    _close(fake, g['g.fake'], 1e-5, 1e-5, 'generated feats (G step)')
    for k in gen:
        _close(_cut(gg[k], full), g[f'g.grad.{k}'], 2e-3, 2e-5, f'generator grad {k}')
        gen[k] = og.adam_step(gen[k], gg[k], {}, 1e-4)
    for k in kg:
        _close(_cut(kg[k], full), g[f'g.grad.kw.{k}'], 2e-4, 2e-6, f'keyword head grad {k}')
        kwp[k] = og.adam_step(kwp[k], kg[k], {}, 1e-4)
    for k in d:
        _close_adam(_cut(d[k], full), g[f'after.d.{k}'], [g[f'c{it}.grad.{k}'] for it in range(n_iters)], n_iters, 1e-5,
                    f'param after step d.{k}')
    for k in gen:
        _close_adam(_cut(gen[k], full), g[f'after.g.{k}'], [g[f'g.grad.{k}']], 1, 1e-4, f'param after step g.{k}')
    for k in kg:
        _close_adam(_cut(kwp[k], full), g[f'after.kw.{k}'], [g[f'g.grad.kw.{k}']], 1, 1e-5, f'param after step kw.{k}')


def test_label_rnn_encoder_matches_reference():
    g = load_golden('label_rnn_tiny')
    for mode in ('max', 'mean'):
        out = og.gru_label_encoder(g['word_emb'], g['ids'], g['mask'], g, pool_mode=mode)
        _close(out, g['pool_' + mode], 1e-5, 1e-6, 'label rnn ' + mode)
