"""Label-dimension sharding (gzsl_text_amd/parallel.py) on CPU: world_size 2 and 3 over gloo, kernels replaced by the
test stand-in (tests/ops_cpu.py).  Every rank must reproduce the UNSHARDED golden logits / loss / gradients of the
reference (shard invariance), in train mode too (the dropout hash is keyed on global coordinates).
Covers both collective layouts: batch-split encoder (B % P == 0: all_gather X||PX, reduce_scatter dX||dPX) and the
replicated-encoder fallback (B % P != 0: all_reduce dX, dPX)."""
import os
import socket
import sys
import traceback

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, name, train, mode, q):
    try:
        sys.path.insert(0, ROOT)
        os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        torch.set_num_threads(1 if world > 4 else 2)
        dist.init_process_group('gloo', rank=rank, world_size=world)
        from tests import helpers
        from tests.conftest import load_golden
        from tests.golden import synth
        from tests.ops_cpu import CpuOps
        if mode == 'gan':
            _gan_variant_sharded(helpers, load_golden, synth, CpuOps())
        else:
            g = load_golden(f"{name}_{'train' if train else 'eval'}")
            case = synth.synth_base(**synth.BASE_CASES[name])
            model = helpers.build_model(case, ops=CpuOps())
            model.enable_label_sharding(own_first=(mode == 'own_first'))
            if mode == 'no_grad':  # forward-only scoring (train_base.py:120-137): no backward, nothing saved
                model.train() if train else model.eval()
                model.set_dropout_seed(synth.DROP_SEED)
                with torch.no_grad():
                    logits, loss = model(*helpers.batch_tensors(case))
                assert not logits.requires_grad and not loss.requires_grad
                helpers.close(logits, g['logits'], 1e-5, 2e-5, 'sharded no_grad logits')
                helpers.close(loss, g['loss'], 1e-5, 1e-5, 'sharded no_grad loss')
            else:
                helpers.check_base_against_golden(model, case, g, train)
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, 'ok'))
    except Exception:
        q.put((rank, traceback.format_exc()))


def _gan_variant_sharded(helpers, load_golden, synth, ops, device='cpu'):
    """ConvLabelAttnGAN.forward (full label list and a label subset, finetune.py:242,271) under label sharding against
    the UNSHARDED golden: logits, loss and the output_fc gradient."""
    from gzsl_text_amd.modules.icd_modules import ConvLabelAttnGAN
    g = load_golden('gan_model_tiny')
    case = synth.synth_base(**synth.BASE_CASES['base_gate_tiny'])
    model = helpers.build_model(case, cls=ConvLabelAttnGAN, device=device, ops=ops)
    model.load_pretrained_state_dict({k: torch.from_numpy(v.copy()).to(device) for k, v in case['params'].items()})
    model.eval()
    model.init_output_fc(torch.from_numpy(g['fc0']).to(device))
    model.enable_label_sharding()
    x, y, mask = helpers.batch_tensors(case, device)
    model.zero_grad()
    logits, loss = model(x, y, mask)
    loss.backward()
    helpers.close(logits, g['full_logits'], 1e-5, 2e-5, 'sharded GAN-variant logits')
    helpers.close(loss, g['full_loss'], 1e-5, 1e-5, 'sharded GAN-variant loss')
    helpers.close(model.output_fc.weight.grad, g['full_dfc'], 1e-4, 1e-6, 'sharded GAN-variant dfc')
    model.zero_grad()
    logits, loss = model(x, y, mask, label_indices=torch.from_numpy(g['sub']).to(device))
    loss.backward()
    helpers.close(logits, g['sub_logits'], 1e-5, 2e-5, 'sharded subset logits')
    helpers.close(loss, g['sub_loss'], 1e-5, 1e-5, 'sharded subset loss')
    helpers.close(model.output_fc.weight.grad, g['sub_dfc'], 1e-4, 1e-6, 'sharded subset dfc')
    with torch.no_grad():
        logits, _ = model(x, y, mask)
    helpers.close(logits, g['full_logits'], 1e-5, 2e-5, 'sharded GAN-variant scoring under no_grad')


def run_ranks(target, world, args, timeout=240):
    """Spawn `world` rank processes running target(rank, world, port, *args, q); every rank must post 'ok' and exit 0.
    Whatever happens (a rank that dies without posting, a timeout, a failed assert) no process is left behind: live
    ranks would otherwise sit in a collective and keep their memory for the rest of the session."""
    import queue
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port) + tuple(args) + (q,)) for r in range(world)]
    res = []
    try:
        for p in procs:
            p.start()
        for _ in procs:
            try:
                res.append(q.get(timeout=timeout))
            except queue.Empty:
                codes = [p.exitcode for p in procs]
                raise AssertionError(f'only {len(res)} of {world} ranks reported within {timeout} s; exit codes {codes}; '
                                     f'reports so far: {res}')
            if res[-1][1] != 'ok':  # a failed rank leaves its peers blocked in a collective: stop waiting for them
                break
        for rank, msg in res:
            assert msg == 'ok', f'rank {rank}:\n{msg}'
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0, f'rank process exit code {p.exitcode}'
    finally:
        for p in procs:
            if p.is_alive():
                p.terminate()
                p.join(timeout=10)
            if p.is_alive():
                p.kill()
                p.join(timeout=10)


def _run(world, name, train, mode='default'):
    run_ranks(_worker, world, (name, train, mode))


@pytest.mark.parametrize('train', [False, True])
def test_label_sharding_world2_replicated_encoder(train):
    _run(2, 'base_gate_tiny', train)  # B = 3: not divisible -> replicated encoder + all_reduce


@pytest.mark.parametrize('train', [False, True])
def test_label_sharding_world2_batch_split(train):
    _run(2, 'base_gate_medium', train)  # B = 2, L = 256: batch-split encoder, all_gather / reduce_scatter


def test_label_sharding_world3_uneven_labels():
    _run(3, 'base_gate_tiny', True)  # L = 12 over 3 ranks, B = 3 batch-split, GAN-free base path


@pytest.mark.parametrize('world,name', [(2, 'base_gate_medium'), (3, 'base_gate_tiny')])
def test_label_sharding_own_notes_first_ordering(world, name):
    _run(world, name, True, mode='own_first')  # own block under the gathers; others first, then own, in backward


@pytest.mark.parametrize('mode', ['default', 'own_first'])
def test_label_sharding_world8(mode):
    _run(8, 'base_gate_w8', True, mode=mode)  # B = 8, L = 19: one note per rank, label ranges of 3 and 2 rows


@pytest.mark.parametrize('world,name,train', [(2, 'base_gate_medium', False), (3, 'base_gate_tiny', True)])
def test_label_sharding_forward_only_under_no_grad(world, name, train):
    _run(world, name, train, mode='no_grad')


@pytest.mark.parametrize('world', [2, 3])
def test_label_sharding_gan_variant(world):
    _run(world, None, False, mode='gan')  # world 2: replicated encoder (B = 3); world 3: batch-split + all-gather


def test_partition_covers_range():
    from gzsl_text_amd.parallel import partition
    for n in (0, 1, 7, 9000, 12):
        for w in (1, 2, 3, 8):
            parts = [partition(n, w, r) for r in range(w)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(parts[i][1] == parts[i + 1][0] for i in range(w - 1))
            sizes = [e - b for b, e in parts]
            assert max(sizes) - min(sizes) <= 1
