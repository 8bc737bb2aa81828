"""Times the GEMM shapes of the WGAN-GP step at a given batch one by one (events on the launch stream).

    python tools/bench_gan_gemm.py [--bg 4096] [--iters 30]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gzsl_text_amd.ops_hip import HipOps  # noqa: E402


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--bg', type=int, default=4096)
    ap.add_argument('--iters', type=int, default=30)
    args = ap.parse_args()
    Bg, D, Ls, H, Vk, E = args.bg, 400, 400, 800, 4096, 200
    ops = HipOps()
    g = torch.Generator(device='cuda').manual_seed(0)

    def rnd(*s):
        return torch.randn(*s, device='cuda', generator=g) * 0.1

    W1, W2, We = rnd(H, D + Ls), rnd(D, H), rnd(Vk, E)
    x, x2, h, lst, F3, lg = rnd(Bg, D), rnd(2 * Bg, D), rnd(Bg, H), rnd(3 * Bg, H), rnd(3 * Bg, D), rnd(Bg, Vk)
    xk, xin = rnd(Bg, E), rnd(Bg, D + Ls)
    cases = [
        ('x @ W1f^T        [Bg,400]x[400,800]', 2 * Bg * D * H, lambda: ops.mm(x, W1[:, :D].t())),
        ('[2Bg] @ W1f^T    [2Bg,400]x[400,800]', 4 * Bg * D * H, lambda: ops.mm(x2, W1[:, :D].t())),
        ('xin @ W1^T       [Bg,800]x[800,800]', 2 * Bg * (D + Ls) * H, lambda: ops.mm(xin, W1.t())),
        ('h @ W2^T         [Bg,800]x[800,400]', 2 * Bg * H * D, lambda: ops.mm(h, W2.t())),
        ('v @ W1f          [Bg,800]x[800,400]', 2 * Bg * H * D, lambda: ops.mm(h, W1[:, :D])),
        ('lst^T @ F        [800,3Bg]x[3Bg,400]', 6 * Bg * H * D, lambda: ops.mm(lst.t(), F3)),
        ('dhs^T @ lab      [800,Bg]x[Bg,400]', 2 * Bg * H * D, lambda: ops.mm(h.t(), x)),
        ('dh^T @ xin       [800,Bg]x[Bg,800]', 2 * Bg * H * (D + Ls), lambda: ops.mm(h.t(), xin)),
        ('xk @ We^T        [Bg,200]x[200,4096]', 2 * Bg * E * Vk, lambda: ops.mm(xk, We.t())),
        ('dlogits @ We     [Bg,4096]x[4096,200]', 2 * Bg * E * Vk, lambda: ops.mm(lg, We)),
    ]
    W1t, W2t, Wet = W1.t().contiguous(), W2.t().contiguous(), We.t().contiguous()
    cases += [
        ('x @ (W1f^T stored [K,N])', 2 * Bg * D * H, lambda: ops.mm(x, W1t[:D])),
        ('[2Bg] @ (W1f^T stored [K,N])', 4 * Bg * D * H, lambda: ops.mm(x2, W1t[:D])),
        ('xin @ (W1^T stored [K,N])', 2 * Bg * (D + Ls) * H, lambda: ops.mm(xin, W1t)),
        ('h @ (W2^T stored [K,N])', 2 * Bg * H * D, lambda: ops.mm(h, W2t)),
        ('xk @ (We^T stored [K,N])', 2 * Bg * E * Vk, lambda: ops.mm(xk, Wet)),
        ('transpose W1 (torch)', 1, lambda: W1.t().contiguous()),
    ]
    tot = 0.0
    for name, fl, fn in cases:
        ms = timeit(fn, args.iters)
        tot += ms
        print(f'{name:40s} {ms * 1e3:8.1f} us  {fl / ms / 1e9:7.2f} TFLOP/s  {fl / ms / 1e9 / 157.3 * 100:5.1f} %')


if __name__ == '__main__':
    main()
