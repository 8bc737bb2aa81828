"""Diagnostic: label attention forward / backward launch times at the bench shape, with and without dropout."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gzsl_text_amd.ops_hip import HipOps  # noqa: E402


def timed(fn, n=int(os.environ.get('N_IT', 5))):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    ops = HipOps()
    B, Tp, E, F, L = int(os.environ.get('B', 16)), 2491, int(os.environ.get('E', 100)), 200, int(os.environ.get('L', 9000))
    g = torch.Generator(device='cuda').manual_seed(0)
    PX = torch.tanh(torch.randn(B, Tp, E, device='cuda', generator=g))
    X = torch.relu(torch.randn(B, Tp, F, device='cuda', generator=g))
    Q = torch.randn(L, E, device='cuda', generator=g) * 0.1
    dO = torch.randn(B, L, F, device='cuda', generator=g) * 0.01
    for p in (0.2, 0.0):
        O, lse = ops.attn_fwd(PX, X, Q, 0, L, 1, p)
        dPX, dX, dQ = torch.zeros_like(PX), torch.zeros_like(X), torch.zeros_like(Q)
        tf = timed(lambda: ops.attn_fwd(PX, X, Q, 0, L, 1, p))
        tb = 0.0 if os.environ.get('FWD_ONLY') else timed(lambda: ops.attn_bwd(dO, O, PX, X, Q, lse, 0, L, 1, p, dPX, dX, dQ))
        print(f'p={p}: fwd {tf:.3f} ms, bwd (delta + main) {tb:.3f} ms', flush=True)


if __name__ == '__main__':
    main()
