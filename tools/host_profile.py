"""Diagnostic: host-side cost of issuing ONE cfg2 training step (queues empty before each step), with a cProfile listing."""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from gzsl_text_amd.modules.common_module import get_optimizer  # noqa: E402
from gzsl_text_amd.utils import synth  # noqa: E402


def main():
    class A:
        config, T, E, L, N, V, batch_per_gpu, weak = 'cfg2', 2500, 100, 9000, 12000, 50001, 16, True
    dev = torch.device('cuda', 0)
    case = bench.workload(A, 1)
    model = synth.build_model(case, device=dev)
    model.train()
    model.set_dropout_seed(1234)
    x, y, mask = synth.batch_tensors(case, dev)
    opt = get_optimizer(1e-3, model, weight_decay=1e-5)

    def step():
        opt.zero_grad(set_to_none=True)
        logits, loss = model(x, y, mask)
        loss.mean().backward()
        opt.step()

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    ts = []
    for _ in range(10):
        t0 = time.perf_counter()
        step()
        ts.append(time.perf_counter() - t0)
        torch.cuda.synchronize()
    print('host issue ms per step (queues empty):', ' '.join(f'{t * 1e3:.2f}' for t in ts))
    pr = cProfile.Profile()
    for _ in range(10):
        pr.enable()
        step()
        pr.disable()
        torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats('tottime').print_stats(30)


if __name__ == '__main__':
    main()
