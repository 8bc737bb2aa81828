"""Diagnostic: time GEMM shapes of the WGAN-GP step and the label encoder under each tile width (GZSL_GEMM_WN=13/7/8;
one process per setting because the library reads the variable once).  python tools/sweep_gemm_wn.py"""
import os
import subprocess
import sys

SHAPES = [(39856, 200, 1000), (39856, 100, 200), (144000, 200, 200), (39856, 200, 100), (8192, 800, 800), (8192, 400, 800),
          (4096, 800, 800), (4096, 400, 800), (4096, 800, 400), (4096, 4096, 200), (4096, 200, 400), (16384, 800, 800),
          (16384, 400, 800), (12000, 300, 100), (12000, 300, 200), (12000, 600, 200), (2048, 800, 800), (1024, 800, 800)]


def child():
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from gzsl_text_amd.ops_hip import HipOps
    ops = HipOps()
    for M, N, K in SHAPES:
        A, W = torch.randn(M, K, device='cuda'), torch.randn(N, K, device='cuda')
        out = torch.empty(M, N, device='cuda')
        for _ in range(3):
            ops.linear(A, W, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.linear(A, W, out=out)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print(f'wn={os.environ.get("GZSL_GEMM_WN", "auto"):>4} {M:6d}x{N:5d}x{K:4d} {ms * 1e3:8.1f} us {2.0 * M * N * K / ms / 1e9:7.1f} TFLOP/s',
              flush=True)


if __name__ == '__main__':
    if os.environ.get('SWEEP_CHILD'):
        child()
    else:
        for wn in ('auto', '13', '7', '8'):
            env = dict(os.environ, SWEEP_CHILD='1')
            if wn != 'auto':
                env['GZSL_GEMM_WN'] = wn
            subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, check=True)
