"""Diagnostic: a few cfg2 training steps with GZSL_ATTN_BF16X6 on, for rocprofv3 --kernel-trace + tools/timeline.py."""
import os
import sys

import torch

os.environ['GZSL_ATTN_BF16X6'] = '1'
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
    assert model._engine().ops.attn_bf16x6
    for _ in range(int(os.environ.get('N_STEPS', 6))):
        opt.zero_grad(set_to_none=True)
        logits, loss = model(x, y, mask)
        loss.mean().backward()
        opt.step()
    torch.cuda.synchronize()
    print('loss', float(loss.detach()))


if __name__ == '__main__':
    main()
