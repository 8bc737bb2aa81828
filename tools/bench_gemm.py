"""Times the GEMM launches of one cfg2 training step one by one (events on the launch stream).

    python tools/bench_gemm.py [--iters 20]

Shapes: BASELINE.json configs[1] (B=16, T=2500, E=H=100, F=200, k=10, L=9000, N=12000 nodes, D=200).
Prints per launch: ms, TFLOP/s and the fraction of the 157.3 TFLOP/s fp32 MFMA peak.
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gzsl_text_amd.ops_hip import HipOps  # noqa: E402
from gzsl_text_amd.hip import ACT_RELU, ACT_TANH  # noqa: E402

PEAK = 157.3


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
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--E', type=int, default=100, help='E = H; D = 2E (200: the width the reference drivers build)')
    args = ap.parse_args()
    dev = 'cuda'
    ops = HipOps()
    B, T, E, F, k, L, N, H, D = 16, 2500, args.E, 200, 10, 9000, 12000, args.E, 2 * args.E
    Tp = T - k + 1
    g = torch.Generator(device=dev).manual_seed(0)

    def rnd(*s):
        return torch.randn(*s, device=dev, generator=g) * 0.1

    e, Wc, bc = rnd(B, T, E), rnd(F, E, k), rnd(F)
    X = rnd(B, Tp, F)
    Wp, bp = rnd(E, F), rnd(E)
    O = rnd(B * L, F)
    Wo, bo = rnd(D, F), rnd(D)
    dDir = rnd(B * L, D)
    dPX = rnd(B * Tp, E)
    dX = rnd(B * Tp, F)
    hN, Wih, dg = rnd(N, H), rnd(3 * H, H), rnd(N, 3 * H)
    rows = []

    def add(name, flops, fn):
        ms = timeit(fn, args.iters)
        tf = flops / (ms * 1e-3) / 1e12
        rows.append((name, ms, tf))
        print(f'{name:34s} {ms*1e3:9.1f} us  {tf:7.2f} TFLOP/s  {tf/PEAK*100:5.1f} %', flush=True)

    M1 = B * Tp
    add('conv fwd      39856x200x1000 KC/KC', 2.0 * M1 * F * E * k, lambda: ops.conv_relu(e, Wc, bc))
    add('conv bwd (dW+de+db)', 2 * 2.0 * M1 * F * E * k, lambda: ops.conv_bwd(X, e, Wc))
    add('proj+tanh     39856x100x200  KC/KC', 2.0 * M1 * E * F, lambda: ops.linear(X.view(M1, F), Wp, bp, act=ACT_TANH))
    add('proj dgrad    39856x200x100  KC/RC', 2.0 * M1 * E * F, lambda: ops.mm(dPX, Wp, out=dX, accumulate=True))
    add('proj wgrad    100x200x39856  RC/RC', 2.0 * M1 * E * F, lambda: ops.mm(dPX.t(), X.view(M1, F)))
    add('outproj fwd   144000x200x200 KC/KC', 2.0 * B * L * F * D, lambda: ops.linear(O, Wo, bo, act=ACT_RELU))
    add('outproj dgrad 144000x200x200 KC/RC', 2.0 * B * L * F * D, lambda: ops.mm(dDir, Wo))
    add('outproj wgrad 200x200x144000 RC/RC', 2.0 * B * L * F * D, lambda: ops.mm(dDir.t(), O))
    add('gru fwd       12000x300x100  KC/KC', 2.0 * N * 3 * H * H, lambda: ops.linear(hN, Wih))
    add('gru dgrad     12000x100x300  KC/RC', 2.0 * N * 3 * H * H, lambda: ops.mm(dg, Wih))
    add('gru wgrad     300x100x12000  RC/RC', 2.0 * N * 3 * H * H, lambda: ops.mm(dg.t(), hN))
    if os.environ.get('GZ_RC_B') == '1':  # forward products with the weight stored [K,N] (B contiguous along its rows)
        Wpt, Wot, Wiht = Wp.t().contiguous(), Wo.t().contiguous(), Wih.t().contiguous()
        A = rnd(M1, E * k)
        Wck, Wckt = Wc.view(F, E * k), Wc.view(F, E * k).t().contiguous()
        add('plain 39856x200x1000  KC/KC', 2.0 * M1 * F * E * k, lambda: ops.mm(A, Wck.t(), bias=bc, act=ACT_RELU))
        add('plain 39856x200x1000  KC/RC', 2.0 * M1 * F * E * k, lambda: ops.mm(A, Wckt, bias=bc, act=ACT_RELU))
        add('proj+tanh  KC/RC', 2.0 * M1 * E * F, lambda: ops.mm(X.view(M1, F), Wpt, bias=bp, act=ACT_TANH))
        add('outproj fwd KC/RC', 2.0 * B * L * F * D, lambda: ops.mm(O, Wot, bias=bo, act=ACT_RELU))
        add('gru fwd KC/RC', 2.0 * N * 3 * H * H, lambda: ops.mm(hN, Wiht))
    if os.environ.get('GZ_TORCH_MM') == '1':  # what the vendor library (rocBLAS / hipBLASLt through torch.mm) does on the same shapes
        add('torch.mm outproj fwd  144000x200x200', 2.0 * B * L * F * D, lambda: torch.mm(O, Wo.t()))
        add('torch.mm outproj dgrad', 2.0 * B * L * F * D, lambda: torch.mm(dDir, Wo))
        add('torch.mm outproj wgrad 200x200x144000', 2.0 * B * L * F * D, lambda: torch.mm(dDir.t(), O))
        A = rnd(M1, E * k)
        add('torch.mm conv-shaped 39856x200x1000', 2.0 * M1 * F * E * k, lambda: torch.mm(A, Wc.view(F, E * k).t()))
        add('torch.mm conv wgrad-shaped 1000x200x39856', 2.0 * M1 * F * E * k, lambda: torch.mm(A.t(), X.view(M1, F)))
        add('torch.mm proj 39856x100x200', 2.0 * M1 * E * F, lambda: torch.mm(X.view(M1, F), Wp.t()))
        add('torch.mm proj dgrad 39856x200x100', 2.0 * M1 * E * F, lambda: torch.mm(dPX, Wp))
    per_step = (rows[0][1] + rows[1][1] + rows[2][1] + rows[3][1] + rows[4][1] + rows[5][1] + rows[6][1] + rows[7][1]
                + 4 * (rows[8][1] + rows[9][1] + rows[10][1]))
    print(f'GEMM family per training step ~ {per_step:.3f} ms')


if __name__ == '__main__':
    main()
