"""Capture golden vectors from the reference implementation (runs ONLY in the build container).

Imports petuum/gzsl_text's `modules/*` from /root/reference on CPU (torch fp32), feeds them the
seeded synthetic inputs of tests/golden/synth.py and writes tests/golden/*.npz.  The reference
never travels to the GPU box; these fixtures (data only) and this script are what is committed.

    python tests/golden/make_golden.py

Train-mode cases replace the two `nn.Dropout` members of the LIVE reference module with a module
that multiplies by a mask produced by the build's counter RNG (oracle/rng.py), so the same seed
reproduces the same mask in the HIP kernels; everything else is the reference's own code.
"""
import os
import sys

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, '/root/reference')

from modules.icd_modules import ConvLabelAttnModel, ConvLabelAttnGAN  # noqa: E402  (reference)
from modules.common_module import get_loss_fn, AdamW  # noqa: E402
from modules.generative_module import (ConditionalGenerator, ConditionalDiscriminator,  # noqa: E402
                                       calc_gradient_penalty)
from modules.decoder_module import load_decoder  # noqa: E402
from modules.lm_module import RNNLabelEncoder  # noqa: E402

from oracle import rng  # noqa: E402
from tests.golden import synth  # noqa: E402

torch.set_num_threads(4)


class InjectedMask(nn.Module):
    """Stands in for nn.Dropout on the live reference module: multiplies by a precomputed mask."""

    def __init__(self, fn):
        super().__init__()
        self.fn = fn

    def forward(self, t):
        return t * self.fn(t.shape)


def build_reference_base(case, cls=ConvLabelAttnModel, **extra):
    d = case['dims']
    bufs = case['bufs']
    rows, cols, vals, N = bufs['adj_matrix']
    adj = torch.from_numpy(synth.coo_to_dense(rows, cols, vals, N))
    model = cls(case['params']['emb.embed.weight'].copy(), torch.from_numpy(bufs['code_idx_matrix']),
                torch.from_numpy(bufs['code_idx_mask']), adj, torch.from_numpy(bufs['num_neighbors']),
                get_loss_fn('bce', reduction='sum'), num_filters=d['F'], kernel_sizes=(d['k'],),
                eval_code_size=d['L'], label_hidden_size=d['H'], graph_encoder=case['encoder'],
                target_count=case['target_count'].copy(), C=2.0, **extra)
    sd = model.state_dict()
    for k, v in case['params'].items():
        assert k in sd and tuple(sd[k].shape) == v.shape, (k, v.shape)
        sd[k] = torch.from_numpy(v.copy())
    model.load_state_dict(sd)
    model.adj_matrix = model.adj_matrix.to_sparse()  # as train_base.py:79 does
    return model


def capture_base(name, kw, train):
    case = synth.synth_base(**kw)
    d = case['dims']
    model = build_reference_base(case)
    x, y, mask = (torch.from_numpy(case[k]) for k in ('x', 'y', 'mask'))
    if train:
        model.train()
        model.emb_drop = InjectedMask(lambda s: torch.from_numpy(
            rng.emb_dropout_scale(synth.DROP_SEED, s[0], s[1], s[2], 0.5)))
        model.attn_drop = InjectedMask(lambda s: torch.from_numpy(
            rng.attn_dropout_scale(synth.DROP_SEED, np.arange(s[0]), np.arange(s[1]), d['L'],
                                   np.arange(s[2]), 0.2)))
    else:
        model.eval()
    model.zero_grad()
    avg, graph = model.get_label_embedding()
    logits, loss = model(x, y, mask)
    loss.mean().backward()
    out = dict(logits=logits.detach().numpy(), loss=np.float32(loss.item()),
               avg=avg.detach().numpy(), graph=graph.detach().numpy(),
               class_margin=model.class_margin.numpy(),
               input_checksum=synth.checksum(*[case['params'][k] for k in sorted(case['params'])],
                                             case['x'], case['y'], case['mask']))
    for k, prm in model.named_parameters():
        out['grad.' + k] = prm.grad.detach().numpy()
    if not train:
        # intermediates used by forward_real_feats consumers (icd_modules.py:152-160)
        with torch.no_grad():
            direct, attn, _ = model.forward_direct_feats(x, avg[:d['L']] if False else avg, mask)
        out['direct'] = direct.numpy()
        out['attn'] = attn.numpy()
    if d['T'] <= 64:  # tiny cases also carry their inputs, so they do not depend on regeneration
        for k, v in case['params'].items():
            out['param.' + k] = v
        out.update(x=case['x'], y=case['y'], mask=case['mask'], target_count=case['target_count'],
                   code_idx_matrix=case['bufs']['code_idx_matrix'], code_idx_mask=case['bufs']['code_idx_mask'],
                   adj_rows=case['bufs']['adj_matrix'][0], adj_cols=case['bufs']['adj_matrix'][1],
                   num_neighbors=case['bufs']['num_neighbors'])
    fn = os.path.join(HERE, f"{name}_{'train' if train else 'eval'}.npz")
    np.savez_compressed(fn, **out)
    print('wrote', fn, f'{os.path.getsize(fn) / 1e6:.2f} MB', 'loss', out['loss'])


def capture_gan_model():
    """ConvLabelAttnGAN: label-subset forward, forward_real_feats, forward_final, AdamW (finetune.py path)."""
    kw = synth.BASE_CASES['base_gate_tiny']
    case = synth.synth_base(**kw)
    d = case['dims']
    model = build_reference_base(case, cls=ConvLabelAttnGAN)
    model.load_pretrained_state_dict({k: torch.from_numpy(v.copy()) for k, v in case['params'].items()})
    model.eval()
    rs = np.random.RandomState(77)
    with torch.no_grad():
        avg, graph = model.get_froze_label_emb()
        fc0 = graph.numpy() + (rs.standard_normal(graph.shape) * 0.05).astype(np.float32)
        model.init_output_fc(torch.from_numpy(fc0))
    x, y, mask = (torch.from_numpy(case[k]) for k in ('x', 'y', 'mask'))
    out = dict(fc0=fc0, frozen_avg=avg.numpy(), frozen_graph=graph.numpy())
    # full-label forward
    model.zero_grad()
    logits, loss = model(x, y, mask)
    loss.backward()
    out.update(full_logits=logits.detach().numpy(), full_loss=np.float32(loss.item()),
               full_dfc=model.output_fc.weight.grad.numpy().copy())
    # label-subset forward (finetune.py:271)
    sub = np.array([7, 2, 9, 4, 11], np.int64)
    model.zero_grad()
    logits, loss = model(x, y, mask, label_indices=torch.from_numpy(sub))
    loss.backward()
    out.update(sub=sub, sub_logits=logits.detach().numpy(), sub_loss=np.float32(loss.item()),
               sub_dfc=model.output_fc.weight.grad.numpy().copy())
    with torch.no_grad():
        direct, attn = model.forward_real_feats(x, mask)
    out.update(real_direct=direct.numpy(), real_attn=attn.numpy())
    # forward_final + AdamW (finetune.py:283-285, common_module.py:138-182)
    Bs = 10
    feats = np.maximum(rs.standard_normal((Bs, d['D'])), 0).astype(np.float32)
    labs = rs.randint(0, d['N'] if False else d['L'], size=Bs).astype(np.int64)
    labs[1] = labs[0]  # duplicate label rows must accumulate
    opt = AdamW(model.output_fc.parameters(), lr=1e-3, weight_decay=5e-4, betas=(0.5, 0.999))
    out.update(ff_feats=feats, ff_labels=labs)
    for step in (1, 2):
        model.zero_grad()
        loss = model.forward_final(torch.from_numpy(feats), torch.from_numpy(labs)) / Bs
        loss.backward()
        out[f'ff_loss{step}'] = np.float32(loss.item())
        out[f'ff_dfc{step}'] = model.output_fc.weight.grad.numpy().copy()
        opt.step()
        out[f'ff_fc_after{step}'] = model.output_fc.weight.detach().numpy().copy()
    fn = os.path.join(HERE, 'gan_model_tiny.npz')
    np.savez_compressed(fn, **out)
    print('wrote', fn, f'{os.path.getsize(fn) / 1e6:.2f} MB')


def capture_gan_model_train(name):
    """ConvLabelAttnGAN.forward in TRAIN mode, as finetune.py:258,271 calls it (model.train(); forward(x, y, mask,
    label_indices=syn_indices)): both dropouts live, replaced on the reference module by the build's counter masks.  The
    attention mask is keyed on row = note * (number of attended labels) + (row of the attended list): for a gathered
    subset the rows are numbered locally, 0 .. |sub|-1 -- the reference draws a fresh mask of that [B, |sub|, T'] shape.
    The first forward (full list) uses seed DROP_SEED, the second (subset) DROP_SEED + 1: one seed per training forward."""
    case, fc0, sub = synth.gan_train_inputs(name)
    model = build_reference_base(case, cls=ConvLabelAttnGAN)
    model.load_pretrained_state_dict({k: torch.from_numpy(v.copy()) for k, v in case['params'].items()})
    model.init_output_fc(torch.from_numpy(fc0))
    model.train()
    seed = [synth.DROP_SEED]
    model.emb_drop = InjectedMask(lambda s: torch.from_numpy(rng.emb_dropout_scale(seed[0], s[0], s[1], s[2], 0.5)))
    model.attn_drop = InjectedMask(lambda s: torch.from_numpy(
        rng.attn_dropout_scale(seed[0], np.arange(s[0]), np.arange(s[1]), s[1], np.arange(s[2]), 0.2)))
    x, y, mask = (torch.from_numpy(case[k]) for k in ('x', 'y', 'mask'))
    out = dict(input_checksum=synth.checksum(*[case['params'][k] for k in sorted(case['params'])], case['x'], case['y'],
                                             case['mask'], fc0, sub))
    model.zero_grad()
    logits, loss = model(x, y, mask)
    loss.backward()
    out.update(full_logits=logits.detach().numpy(), full_loss=np.float32(loss.item()),
               full_dfc=model.output_fc.weight.grad.numpy().copy())
    seed[0] = synth.DROP_SEED + 1
    model.zero_grad()
    logits, loss = model(x, y, mask, label_indices=torch.from_numpy(sub))
    loss.backward()
    dfc = model.output_fc.weight.grad.numpy()
    assert not np.delete(dfc, sub, axis=0).any()   # only the subset's rows receive gradient
    out.update(sub_logits=logits.detach().numpy(), sub_loss=np.float32(loss.item()), sub_dfc_rows=dfc[sub].copy())
    fn = os.path.join(HERE, f'gan_model_train_{name}.npz')
    np.savez_compressed(fn, **out)
    print('wrote', fn, f'{os.path.getsize(fn) / 1e6:.2f} MB', 'full', out['full_loss'], 'sub', out['sub_loss'])


def capture_gan_step(name, seed, n_iters, store_inputs, **dims):
    """One WGAN-GP step in the call order of train_gan.py:223-308 (critic x n_iters, generator x 1)."""
    s = synth.synth_gan(seed, n_iters=n_iters, **dims)
    dd = s['dims']
    G = ConditionalGenerator(dd['Ls'], dd['Ls'], dd['ngh'], dd['D'])
    Dn = ConditionalDiscriminator(dd['D'], dd['Ls'], dd['ndh'], 1)
    G.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in s['g'].items()})
    Dn.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in s['d'].items()})
    KW = load_decoder('linear', dd['D'], dd['E'], torch.from_numpy(s['kw']['word_emb'].copy()))
    KW.proj_fc.weight.data.copy_(torch.from_numpy(s['kw']['proj_fc.weight']))
    KW.proj_fc.bias.data.copy_(torch.from_numpy(s['kw']['proj_fc.bias']))
    lr, reg_ratio = 1e-4, 0.01
    opt_d = torch.optim.Adam(Dn.parameters(), lr=lr, betas=(0.5, 0.999))
    opt_g = torch.optim.Adam(list(G.parameters()) + list(KW.parameters()), lr=lr, betas=(0.5, 0.999))
    one = torch.ones([])
    mone = one * -1
    out = dict(input_checksum=synth.checksum(*[s['g'][k] for k in sorted(s['g'])],
                                             *[s['d'][k] for k in sorted(s['d'])],
                                             s['iters'][0]['real'], s['iters'][-1]['noise']))
    real_rand = torch.rand
    kw_idx_t, kw_mask_t = torch.from_numpy(s['kw_idx']), torch.from_numpy(s['kw_mask'])

    def generator_iteration(tag):
        """train_gan.py:262-306 up to (not including) optimizer_g.step(); dumps costs, fake and gradients under `tag`."""
        for p in Dn.parameters():
            p.requires_grad = False
        I = s['iters'][n_iters]
        lab, noise, sim = (torch.from_numpy(I[k]) for k in ('lab', 'noise', 'sim'))
        opt_g.zero_grad()
        fake = torch.relu(G(noise, lab))
        kw_loss = KW(fake, kw_idx_t, kw_mask_t, lab)
        kww = sim.masked_fill(sim != 1, 0.)
        kw_loss = torch.mean(kw_loss * kww)
        fake_logits = Dn(fake, lab)
        g_cost = -(fake_logits * sim).mean()
        (g_cost + reg_ratio * kw_loss).backward()
        out[f'{tag}.g_cost'] = np.float32(g_cost.item())
        out[f'{tag}.kw_loss'] = np.float32(kw_loss.item())
        out[f'{tag}.fake'] = fake.detach().numpy().copy()
        for k, p in list(G.named_parameters()) + [('kw.' + k, p) for k, p in KW.named_parameters()]:
            g_ = p.grad.numpy()
            out[f'{tag}.grad.{k}'] = g_.copy() if store_inputs else g_.reshape(g_.shape[0], -1)[:4].copy()

    # The generator iteration at the INITIAL (fixed) critic, before any optimizer_d.step(): no Adam-perturbed
    # parameters upstream, so generator / keyword-head gradients can be held to the plain fp32 tolerance.  Nothing is
    # stepped here; the step proper below starts from the same parameters.
    generator_iteration('g0')
    opt_g.zero_grad()

    for it in range(n_iters):
        I = s['iters'][it]
        real, lab, noise, sim = (torch.from_numpy(I[k]) for k in ('real', 'lab', 'noise', 'sim'))
        for p in Dn.parameters():
            p.requires_grad = True
        opt_d.zero_grad()
        real_logits = Dn(real, lab)
        critic_real = (real_logits * sim).mean()
        critic_real.backward(mone, retain_graph=True)
        fake = torch.relu(G(noise, lab))
        fake_logits = Dn(fake.detach(), lab)
        critic_fake = (fake_logits * sim).mean()
        critic_fake.backward(one, retain_graph=True)
        torch.rand = lambda *a, **k: torch.from_numpy(I['alpha'].copy())  # supply alpha (generative_module.py:24)
        try:
            gp = calc_gradient_penalty(Dn, real, fake.data, lab)
        finally:
            torch.rand = real_rand
        gp.backward()
        out[f'c{it}.d_cost'] = np.float32((critic_fake - critic_real).item())
        out[f'c{it}.gp'] = np.float32(gp.item())
        if store_inputs or it == 0:
            out[f'c{it}.fake'] = fake.detach().numpy().copy()
        for k, p in Dn.named_parameters():
            g_ = p.grad.numpy()
            out[f'c{it}.grad.{k}'] = g_.copy() if store_inputs else g_.reshape(g_.shape[0], -1)[:4].copy()
        opt_d.step()
    generator_iteration('g')
    opt_g.step()
    for tag, mod in (('d', Dn), ('g', G), ('kw', KW)):
        for k, p in mod.named_parameters():
            v = p.detach().numpy()
            out[f'after.{tag}.{k}'] = v.copy() if store_inputs else v.reshape(v.shape[0], -1)[:4].copy()
    fn = os.path.join(HERE, f'{name}.npz')
    np.savez_compressed(fn, **out)
    print('wrote', fn, f'{os.path.getsize(fn) / 1e6:.2f} MB', 'gp', out['c0.gp'], 'g_cost', out['g.g_cost'], 'g0', out['g0.g_cost'])


def capture_label_rnn():
    """RNNLabelEncoder.forward_enc(pool 'max' and 'mean') as called by train_gan.py:179-183."""
    rs = np.random.RandomState(5)
    V, E, n, Td = 60, 8, 11, 7
    W = (rs.standard_normal((V, E)) * 0.5).astype(np.float32)
    W[0] = 0
    enc = RNNLabelEncoder(W.copy())
    r = {}
    for k, p in enc.rnn.named_parameters():
        v = rs.uniform(-0.4, 0.4, size=tuple(p.shape)).astype(np.float32)
        p.data.copy_(torch.from_numpy(v))
        r['rnn.' + k] = v
    lens = rs.randint(1, Td + 1, size=n)
    ids = rs.randint(1, V, size=(n, Td)).astype(np.int64)
    m = (np.arange(Td)[None] < lens[:, None]).astype(np.float32)
    ids = ids * m.astype(np.int64)
    out = dict(word_emb=W, ids=ids, mask=m, **r)
    with torch.no_grad():
        for mode in ('max', 'mean'):
            out['pool_' + mode] = enc.forward_enc(torch.from_numpy(ids), torch.from_numpy(m), None,
                                                  training=False, pool_mode=mode).numpy()
    fn = os.path.join(HERE, 'label_rnn_tiny.npz')
    np.savez_compressed(fn, **out)
    print('wrote', fn)


if __name__ == '__main__':
    torch.manual_seed(0)
    only = sys.argv[1:]   # `make_golden.py base_gate_e200` captures just the named base cases
    if only == ['gan_model_train']:
        for name in synth.GAN_TRAIN_CASES:
            capture_gan_model_train(name)
        sys.exit(0)
    for name, kw in synth.BASE_CASES.items():
        if only and name not in only:
            continue
        capture_base(name, kw, train=False)
        capture_base(name, kw, train=True)
    if only and 'gan_step' not in only:
        sys.exit(0)
    if not only:
        capture_gan_model()
        for name in synth.GAN_TRAIN_CASES:
            capture_gan_model_train(name)
    capture_gan_step('gan_step_tiny', seed=11, n_iters=5, store_inputs=True,
                     Bg=6, D=16, Ls=16, ngh=24, ndh=20, E=8, Vk=30, top_k=5)
    capture_gan_step('gan_step_medium', seed=12, n_iters=2, store_inputs=False,
                     Bg=96, D=400, Ls=400, ngh=800, ndh=800, E=200, Vk=300, top_k=20)
    if not only:
        capture_label_rnn()
