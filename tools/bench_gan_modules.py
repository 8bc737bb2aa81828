"""WGAN-GP step through the drop-in nn.Module surface (ConditionalGenerator / ConditionalDiscriminator /
calc_gradient_penalty / LinearKeywordsDecoder + torch autograd + torch.optim.Adam), in train_gan.py:223-308's call
order -- what the unchanged driver runs -- next to the fused engine (tools/bench_gan.py).

    python tools/bench_gan_modules.py [--bg 8] [--steps 20]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gzsl_text_amd.modules.decoder_module import load_decoder  # noqa: E402
from gzsl_text_amd.modules.generative_module import (ConditionalDiscriminator, ConditionalGenerator,  # noqa: E402
                                                     calc_gradient_penalty)
from gzsl_text_amd.utils import synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--bg', type=int, default=8)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--reg-ratio', type=float, default=0.1)
    args = ap.parse_args()
    dev = torch.device('cuda', 0)
    s = synth.synth_gan(seed=7, Bg=args.bg, D=400, Ls=400, ngh=800, ndh=800, E=200, Vk=4096, top_k=20, n_iters=5)
    G = ConditionalGenerator(400, 400, 800, 400)
    Dn = ConditionalDiscriminator(400, 400, 800, 1)
    G.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in s['g'].items()})
    Dn.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in s['d'].items()})
    KW = load_decoder('linear', 400, 200, torch.from_numpy(s['kw']['word_emb'].copy()))
    G, Dn, KW = G.to(dev), Dn.to(dev), KW.to(dev)
    opt_d = torch.optim.Adam(Dn.parameters(), lr=1e-4, betas=(0.5, 0.999))
    opt_g = torch.optim.Adam(list(G.parameters()) + list(KW.parameters()), lr=1e-4, betas=(0.5, 0.999))
    its = [{k: torch.from_numpy(v).to(dev) for k, v in it.items()} for it in s['iters']]
    kw_idx, kw_mask = torch.from_numpy(s['kw_idx']).to(dev), torch.from_numpy(s['kw_mask']).to(dev)
    one = torch.ones([], device=dev)

    def step():
        for p in Dn.parameters():
            p.requires_grad = True
        for i in range(5):
            I = its[i]
            Dn.zero_grad()
            critic_real = (Dn(I['real'], I['lab']) * I['sim']).mean()
            critic_real.backward(-one)
            fake = torch.relu(G(I['noise'], I['lab']))
            critic_fake = (Dn(fake.detach(), I['lab']) * I['sim']).mean()
            critic_fake.backward(one)
            gp = calc_gradient_penalty(Dn, I['real'], fake.data, I['lab'])
            gp.backward()
            opt_d.step()
        for p in Dn.parameters():
            p.requires_grad = False
        I = its[5]
        G.zero_grad()
        KW.zero_grad()
        fake = torch.relu(G(I['noise'], I['lab']))
        kl = torch.mean(KW(fake, kw_idx, kw_mask, I['lab']) * I['sim'].masked_fill(I['sim'] != 1, 0.))
        g_cost = -(Dn(fake, I['lab']) * I['sim']).mean()
        (g_cost + args.reg_ratio * kl).backward()
        opt_g.step()
        return g_cost

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        g_cost = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    assert bool(torch.isfinite(g_cost))
    print(json.dumps({'metric': 'WGAN-GP samples/sec through the nn.Module surface (5 critic + 1 generator iteration, Adam)',
                      'value': args.bg / dt, 'unit': 'samples/s', 'ms_per_step': dt * 1e3, 'Bg': args.bg,
                      'frac_of_157.3': 7.8e7 * args.bg / dt / 1e12 / 157.3}))


if __name__ == '__main__':
    main()
