"""Secondary benchmark: samples/sec of one full WGAN-GP step = 5 critic iterations + 1 generator
iteration with Adam (train_gan.py:223-308), on the HIP engine, inputs resident in HBM.

    python tools/bench_gan.py [--bg 4096] [--steps 10] [--reg-ratio 0.1]

Shapes: D=400 features, Ls=400 label embedding (graph 200 + GRU 200... the drivers' real sizes), noise=400, ngh=ndh=800,
keyword head E=200, Vk=4096, top_k=20.  Algorithmic FLOPs per sample per step: 7.8e7.
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gzsl_text_amd.gan_engine import WganGpStep, graphed_step  # noqa: E402
from gzsl_text_amd.ops_hip import HipOps  # noqa: E402
from gzsl_text_amd.utils import synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--bg', type=int, default=4096)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--reg-ratio', type=float, default=0.1)
    ap.add_argument('--per-iteration', action='store_true', help='critic_step x 5 + generator_step instead of full_step')
    ap.add_argument('--graph', action='store_true', help='replay the step as one HIP graph (small batches are launch-bound)')
    args = ap.parse_args()
    dev = torch.device('cuda', 0)
    case = synth.synth_gan(seed=7, Bg=args.bg, D=400, Ls=400, ngh=800, ndh=800, E=200, Vk=4096, top_k=20, n_iters=5)

    def dv(d):
        return {k: torch.from_numpy(v).to(dev) for k, v in d.items()}

    step = WganGpStep(HipOps(), dv(case['g']), dv(case['d']), dv(case['kw']), lr=1e-4, critic_iters=5,
                      reg_ratio=args.reg_ratio, capturable=args.graph)
    its = [{k: torch.from_numpy(v).to(dev) for k, v in it.items()} for it in case['iters']]
    kw_idx, kw_mask = torch.from_numpy(case['kw_idx']).to(dev), torch.from_numpy(case['kw_mask']).to(dev)

    def one():
        if args.per_iteration:
            for i in range(5):
                it = its[i]
                step.critic_step(it['real'], it['lab'], it['noise'], it['alpha'], it['sim'])
            it = its[5]
            return step.generator_step(it['lab'], it['noise'], it['sim'], kw_idx, kw_mask)[0]
        cb = [(it['real'], it['lab'], it['noise'], it['alpha'], it['sim']) for it in its[:5]]
        it = its[5]
        return step.full_step(cb, (it['lab'], it['noise'], it['sim'], kw_idx, kw_mask))[1][0]

    if args.graph:
        one = graphed_step(one)
    for _ in range(args.warmup):
        one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        g_cost = one()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    assert bool(torch.isfinite(g_cost))
    flops = 7.8e7 * args.bg
    print(json.dumps({'metric': 'WGAN-GP samples/sec (5 critic + 1 generator iteration, Adam)', 'value': args.bg / dt,
                      'unit': 'samples/s', 'ms_per_step': dt * 1e3, 'Bg': args.bg, 'reg_ratio': args.reg_ratio, 'hip_graph': bool(args.graph),
                      'algorithmic_tflops': flops / dt / 1e12, 'frac_of_157.3': flops / dt / 1e12 / 157.3,
                      'dtype': 'f32', 'data': 'synthetic'}))


if __name__ == '__main__':
    main()
