"""Diagnostic: time label_attn_bwd at the bench shape with sections of the kernel compiled out (GZ_ABL bit mask).

Build the variants first (CPU box):  python tools/ablate_bwd.py --build
Then on the GPU box:                  python tools/ablate_bwd.py
Each variant runs in its own process (the library path is read at import).  Outputs are wrong by construction;
only the times are meaningful.
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VARIANTS = {0: 'full', 1: 'no dQ', 2: 'no dK/dV mfma', 4: 'no rng', 8: 'no streaming', 16: 'no epilogue', 32: 'no S/dP',
            3: 'no dQ, no dK/dV', 35: 'S,dP off, dQ off, dKV off', 9: 'no dQ no streaming'}


def build():
    csrc = os.path.join(ROOT, 'gzsl_text_amd', 'csrc')
    out = os.path.join(ROOT, 'tools', '_abl')
    os.makedirs(out, exist_ok=True)
    procs = []
    for v in VARIANTS:
        cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', f'-DGZ_ABL={v}',
               os.path.join(csrc, 'attn.hip'), os.path.join(csrc, 'gemm.hip'), os.path.join(csrc, 'elementwise.hip'),
               os.path.join(csrc, 'gan.hip'), '-o', os.path.join(out, f'lib_abl{v}.so')]
        procs.append(subprocess.Popen(cmd, stderr=subprocess.DEVNULL))
        if len(procs) >= 4:
            for p in procs:
                p.wait()
            procs = []
    for p in procs:
        p.wait()


def run_one():
    import torch
    sys.path.insert(0, ROOT)
    from gzsl_text_amd.ops_hip import HipOps
    ops = HipOps()
    B, Tp, E, F, L = 16, 2491, 100, 200, 9000
    g = torch.Generator(device='cuda').manual_seed(0)
    PX = torch.tanh(torch.randn(B, Tp, E, device='cuda', generator=g))
    X = torch.relu(torch.randn(B, Tp, F, device='cuda', generator=g))
    Q = torch.randn(L, E, device='cuda', generator=g) * 0.1
    O, lse = ops.attn_fwd(PX, X, Q, 0, L, 1, 0.2)
    dO = torch.randn(B, L, F, device='cuda', generator=g) * 0.01
    dPX, dX, dQ = torch.zeros_like(PX), torch.zeros_like(X), torch.zeros_like(Q)
    for nsplit in (4,):
        for _ in range(2):
            ops.attn_bwd(dO, O, PX, X, Q, lse, 0, L, 1, 0.2, dPX, dX, dQ, nsplit=nsplit)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            ops.attn_bwd(dO, O, PX, X, Q, lse, 0, L, 1, 0.2, dPX, dX, dQ, nsplit=nsplit)
        e1.record()
        torch.cuda.synchronize()
        print(f'{os.environ.get("GZ_ABL_NAME", "?"):32s} nsplit={nsplit} {e0.elapsed_time(e1) / 3:8.3f} ms', flush=True)


if __name__ == '__main__':
    if '--build' in sys.argv:
        build()
    elif '--one' in sys.argv:
        run_one()
    else:
        for v, name in VARIANTS.items():
            env = dict(os.environ, GZSL_HIP_LIB=os.path.join(ROOT, 'tools', '_abl', f'lib_abl{v}.so'), GZ_ABL_NAME=f'{v}: {name}')
            subprocess.run([sys.executable, os.path.abspath(__file__), '--one'], env=env, timeout=300)
