"""Diagnostic: where a step of label_attn_bwd_bf16_kernel spends its time, per section, by shader-clock timers.

Build first (here, on the CPU):
    hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -shared -DBB_PROF gzsl_text_amd/csrc/attn_bwd_bf16.hip -o tools/_diag/libbwd_prof.so
then on the GPU box:  python tools/prof_bwd_phases.py
Sections (average shader clocks per step): K role 0 S, 1 softmax + Pd planes, 2 dQ k-step 0, 3 wait X, 4 dS + dK,
5 dQ k-step 1 + stores, 6 wait B, 7 commit + issue, 8 wait A;  V role 0 dP, 3 wait X, 4 dV, 6 wait B, 7 commit + issue, 8 wait A.
"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gzsl_text_amd.ops_hip import HipOps, drop_params  # noqa: E402


def main():
    lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), '_diag', os.environ.get('PROF_LIB', 'libbwd_prof.so')))
    vp, ci, ll = ctypes.c_void_p, ctypes.c_int, ctypes.c_longlong
    lib.gzsl_label_attn_bwd_bf16x6_work_bytes.restype = ll
    lib.gzsl_label_attn_bwd_bf16x6_work_bytes.argtypes = [ci] * 5
    lib.gzsl_label_attn_bwd_bf16x6.argtypes = [vp] * 9 + [ci] * 8 + [ctypes.c_uint32, ctypes.c_uint32, ctypes.c_float, vp, ll, ci, vp]
    ops = HipOps()
    B, Tp, E, F, L = 16, 2491, 100, 200, 9000
    g = torch.Generator(device='cuda').manual_seed(0)
    PX = torch.tanh(torch.randn(B, Tp, E, device='cuda', generator=g))
    X = torch.relu(torch.randn(B, Tp, F, device='cuda', generator=g))
    Q = torch.randn(L, E, device='cuda', generator=g) * 0.1
    dO = torch.randn(B, L, F, device='cuda', generator=g) * 0.01
    ns = int(os.environ.get('NS', 2))
    nwg = B * ((Tp + 63) // 64) * ns
    for p in (0.2, 0.0):
        O, lse = ops.attn_fwd(PX, X, Q, 0, L, 1, p)
        delta = (dO * O).sum(-1)
        dPX, dX, dQ = torch.zeros_like(PX), torch.zeros_like(X), torch.zeros_like(Q)
        need = lib.gzsl_label_attn_bwd_bf16x6_work_bytes(B, Tp, E, F, L)
        extra = nwg * 8 * 10 * 8
        work = torch.zeros(need + extra, dtype=torch.uint8, device='cuda')
        th, sc = drop_params(p)
        st = torch.cuda.current_stream().cuda_stream
        for _ in range(2):
            rc = lib.gzsl_label_attn_bwd_bf16x6(dO.data_ptr(), PX.data_ptr(), X.data_ptr(), Q.data_ptr(), lse.data_ptr(),
                                                delta.data_ptr(), dPX.data_ptr(), dX.data_ptr(), dQ.data_ptr(), B, Tp, E, F, L, 0, L,
                                                ns, 1, th, sc, work.data_ptr(), need + extra, 1, st)
            assert rc == 0, rc
        torch.cuda.synchronize()
        d = work[need:].view(torch.int64).view(nwg, 8, 10).double().cpu()
        steps = ((L + 31) // 32 + ns - 1) // ns
        print(f'p={p}: {nwg} workgroups x {steps} steps; average clocks per step by section')
        for w, name in ((0, 'K tg0'), (3, 'K tg3'), (4, 'V tg0'), (7, 'V tg3')):
            m = d[:, w, :].mean(0) / steps
            print(f'  {name}: ' + ' '.join(f'{k}:{m[k]:7.0f}' for k in range(9)) + f'  sum {m.sum():7.0f}')


if __name__ == '__main__':
    main()
