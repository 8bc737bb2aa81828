"""Label-dimension sharding with the HIP kernels: two ranks share cuda:0 and talk over gloo (RCCL needs one GPU per
rank; the driver's 8-GPU run is the RCCL path).  Every rank must reproduce the UNSHARDED golden logits / loss /
gradients of the reference — this pins what the CPU gloo tests cannot: the kernels' global label coordinates
(l_offset / L_total of the dropout hash), the all-gathered X||PX layout and the reduce-scattered gradients on device."""
import os
import sys
import traceback

import pytest
import torch
import torch.distributed as dist

from tests.test_sharding_cpu import run_ranks

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _worker_nccl(rank, world, port, name, train, own_first, q):
    _worker(rank, world, port, name, train, own_first, q, backend='nccl')


def _worker(rank, world, port, name, train, own_first, q, backend='gloo'):
    try:
        sys.path.insert(0, ROOT)
        os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        torch.cuda.set_device(0)
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', 0))
        else:
            dist.init_process_group('gloo', rank=rank, world_size=world)
        from tests import helpers
        from tests.conftest import load_golden
        from tests.golden import synth
        if name != 'gan_variant':
            g = load_golden(f"{name}_{'train' if train else 'eval'}")
            case = synth.synth_base(**synth.BASE_CASES[name])
        if name == 'gan_variant':
            from gzsl_text_amd.ops_hip import HipOps
            from tests.test_sharding_cpu import _gan_variant_sharded
            _gan_variant_sharded(helpers, load_golden, synth, HipOps(), device='cuda')
        else:
            model = helpers.build_model(case, device='cuda')
            model.enable_label_sharding(own_first=own_first)
            helpers.check_base_against_golden(model, case, g, train, device='cuda', grad_atol=4e-6)
        torch.cuda.synchronize()
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, 'ok'))
    except Exception:
        q.put((rank, traceback.format_exc()))


@pytest.mark.parametrize('name,world,train,own_first', [
    ('base_gate_medium', 2, True, False), ('base_gate_tiny', 3, True, False), ('base_gate_tiny', 2, False, False),
    ('base_gate_e200', 2, True, False), ('base_gate_medium', 2, True, True), ('base_gate_w8', 8, True, True),
    ('gan_variant', 3, False, False)])
def test_hip_label_sharding_matches_unsharded_golden(name, world, train, own_first):
    run_ranks(_worker, world, (name, train, own_first), timeout=200)


@pytest.mark.parametrize('train', [True, False])
def test_rccl_backend_single_rank(train):
    """The code path the 8-GPU run takes ('nccl' = RCCL: device tensors handed to the collectives directly, all-gathers
    and reduce-scatters started with async_op=True and waited on, second-stream label encoder beside them), on the one
    GPU a test box has: a process group of ONE rank.  Catches misuse of the RCCL API (layouts, dtypes, handles, stream
    ordering); the multi-rank arithmetic is pinned by the gloo tests above."""
    run_ranks(_worker_nccl, 1, ('base_gate_medium', train, False), timeout=200)
