import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gzsl_text_amd.ops_hip import HipOps
ops = HipOps()
for M, N, K in [(39856, 200, 1000), (144000, 200, 200), (4096, 800, 800)]:
    A, W = torch.randn(M, K, device='cuda'), torch.randn(N, K, device='cuda')
    out = torch.empty(M, N, device='cuda')
    for _ in range(3): ops.linear(A, W, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.linear(A, W, out=out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f'{os.environ.get("TAG","base"):>8} {M:6d}x{N:5d}x{K:4d} {ms*1e3:8.1f} us {2.0*M*N*K/ms/1e9:7.1f} TFLOP/s', flush=True)
