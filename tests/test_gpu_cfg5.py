"""BASELINE configs[4] at its own shapes (SURVEY.md 8d "Configs -> shapes": cfg5) on the GPU against the numpy oracle:

  * one finetune "epoch" of finetune.py:256-292 -- model.train(); neg_iters = 3 real-note batches through
    `ConvLabelAttnGAN.forward(x, y, mask, label_indices=syn_indices)` (|syn| = 500 of L = 9000 labels, E = H = 200,
    D = 400, B = 2 notes of T = 2000 tokens, both dropouts live) each followed by `AdamW.step()`; one synthetic-feature
    batch through `forward_final` (Bs = 128) / Bs + AdamW; the reset of every non-syn row of `output_fc` to the
    pretrained table;
  * forward-only scoring of the full label list (finetune.py:224-253 / train_base.py:231-250: eval mode, no_grad)
    label-sharded over two ranks that share cuda:0 (gloo; each rank owns 4500 labels and encodes half of the notes).

The oracle (oracle/base_model.py, pinned to the reference by tests/test_oracle_golden.py, train-mode subset forward
included) finishes these sizes in seconds: the subset is 500 labels, the full list is chunked.
Tolerances (fp32): logits / loss rtol 1e-5, gradients rtol 1e-4; output_fc after the AdamW steps: `_close_adamw`.
"""
import os
import sys
import traceback

import numpy as np
import pytest
import torch
import torch.distributed as dist

from oracle import base_model as om
from tests import helpers
from tests.golden import synth
from tests.helpers import close
from tests.test_sharding_cpu import run_ranks

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = 'cuda'
L, N, NSYN, BS = 9000, 9600, 500, 128


def _close_adamw(a, b, grads, lr, what, eps=1e-6, beta1=0.5, grad_noise=4e-7, grad_floor=1e-3):
    """Parameters after len(grads) steps of the no-bias-correction AdamW (common_module.py:138-182),
    p -= lr * (m / (sqrt(v) + eps) + wd * p).  A gradient perturbation d moves one step's update by at most
    lr * (1 - beta1) * d / (sqrt(v) + eps) <= lr * (1 - beta1) * d / eps: where |g| is below ~eps / sqrt(1 - beta2) the
    update is LINEAR in g with slope lr / (2 eps) = 5, so fp32 rounding noise of the gradient (<= grad_noise absolute: a
    B = 2 term sum of dlogit * relu(direct), direct carrying ~1e-6 of accumulation noise) shows up 5x in the parameter.
    Every element: within 1e-6 + 1e-5 |b| + steps * lr * (1 - beta1) / eps * grad_noise.  Elements whose oracle gradient
    exceeded grad_floor in every step so far (sqrt(v) >> eps there): tight, 1e-6 + 1e-5 |b|."""
    a = np.asarray(a.detach().cpu().numpy() if torch.is_tensor(a) else a, np.float64)
    b = np.asarray(b, np.float64)
    err = np.abs(a - b)
    loose = 1e-6 + 1e-5 * np.abs(b) + len(grads) * lr * (1 - beta1) / eps * grad_noise
    assert not (err > loose).any(), f'{what}: max err {err.max():.3e}, {int((err > loose).sum())} elements outside'
    big = np.ones(b.shape, bool)
    for g_ in grads:
        big &= np.abs(g_) > grad_floor
    bad = big & (err > 1e-6 + 1e-5 * np.abs(b))
    assert not bad.any(), f'{what}: {int(bad.sum())} of {int(big.sum())} elements with |grad| > {grad_floor} not tight'


def _case(B, seed):
    return synth.synth_base(seed=seed, B=B, T=2000, E=200, H=200, L=L, N=N, Td=30, V=3000, F=200, labels_per_note=15,
                            emb_scale=0.1, zipf=True)


def _gan_model(case, device):
    from gzsl_text_amd.modules.icd_modules import ConvLabelAttnGAN
    model = helpers.build_model(case, cls=ConvLabelAttnGAN, device=device)
    model.load_pretrained_state_dict({k: torch.from_numpy(v.copy()).to(device) for k, v in case['params'].items()})
    return model


def test_finetune_epoch_at_cfg5_shapes_vs_oracle():
    from gzsl_text_amd.modules.common_module import AdamW
    neg_iters, lr, wd = 3, 1e-5, 5e-4          # config.py:66-97 defaults (lr 1e-5, l2 5e-4, neg_iters 3)
    case = _case(2 * neg_iters, seed=55)
    d = case['dims']
    rs = np.random.RandomState(56)
    syn = rs.choice(L, size=NSYN, replace=False).astype(np.int64)      # few/zero-shot codes, unsorted
    syn_x = np.maximum(rs.standard_normal((BS, d['D'])), 0).astype(np.float32) * 0.3
    syn_y = syn[rs.randint(0, NSYN, size=BS)]                           # generated features' labels (duplicates)
    p = dict(case['params'])
    margin = om.class_margin_from_count(case['target_count'])
    avg, graph, _ = om.label_embedding_fwd(p, case['bufs'])

    model = _gan_model(case, DEV)
    _, pretrain = model.get_froze_label_emb()
    close(pretrain, graph, 1e-5, 2e-6, 'pretrained classifier table (graph label embedding)')
    model.init_output_fc(pretrain)                                      # finetune.py:142
    opt = AdamW(model.output_fc.parameters(), lr=lr, weight_decay=wd, betas=(0.5, 0.999))
    fc = graph.copy()
    m, v = np.zeros_like(fc), np.zeros_like(fc)
    x_all, y_all, mask_all = helpers.batch_tensors(case, DEV)
    syn_t = torch.from_numpy(syn).to(DEV)
    grads = []

    model.train()
    model.set_dropout_seed(synth.DROP_SEED)
    for it in range(neg_iters):                                         # finetune.py:265-273
        sl = slice(2 * it, 2 * it + 2)
        opt.zero_grad()
        logits, loss = model.forward(x_all[sl], y_all[sl], mask_all[sl], label_indices=syn_t)
        loss.backward()
        p['output_fc.weight'] = fc
        want = om.gan_model_forward_backward(p, avg, case['x'][sl], case['y'][sl], case['mask'][sl], L, margin, 2.0,
                                             label_indices=syn, drop_seed=synth.DROP_SEED + it)
        close(logits, want['logits'], 1e-5, 2e-5, f'subset logits, real batch {it}')
        close(loss, want['loss'], 1e-5, 1e-5, f'subset loss, real batch {it}')
        close(model.output_fc.weight.grad[syn_t], want['dfc'][syn], 1e-4, 2e-6, f'output_fc gradient rows, real batch {it}')
        opt.step()
        fc, m, v = om.adamw_nobias_step(fc, want['dfc'], m, v, lr, 0.5, 0.999, 1e-6, wd)
        grads.append(want['dfc'])
        _close_adamw(model.output_fc.weight, fc, grads, lr, f'output_fc after AdamW step {it}')
    opt.zero_grad()                                                     # finetune.py:282-286
    floss = model.forward_final(torch.from_numpy(syn_x).to(DEV), torch.from_numpy(syn_y).to(DEV)) / BS
    floss.backward()
    wl, wdfc, _ = om.forward_final(fc, syn_x, syn_y, margin, 2.0)
    close(floss, wl / BS, 1e-5, 1e-5, 'forward_final loss at Bs=128, D=400')
    close(model.output_fc.weight.grad[syn_t], (wdfc / np.float32(BS))[syn], 1e-4, 1e-6, 'forward_final gradient rows')
    opt.step()
    fc, m, v = om.adamw_nobias_step(fc, wdfc / np.float32(BS), m, v, lr, 0.5, 0.999, 1e-6, wd)
    grads.append(wdfc / np.float32(BS))
    # finetune.py:289-292: every row outside syn goes back to the pretrained table (weight decay moved all of them)
    assert float((model.output_fc.weight.data[:L] - pretrain[:L]).abs().max()) > 0
    temp = model.output_fc.weight.data.clone()
    model.output_fc.weight.data.copy_(pretrain.data)
    model.output_fc.weight.data[syn_t] = temp[syn_t]
    reset = graph.copy()
    reset[syn] = fc[syn]
    _close_adamw(model.output_fc.weight, reset, grads, lr, 'output_fc after the epoch (syn rows trained, the rest reset)')
    rest = np.ones(N, bool)
    rest[syn] = False
    assert torch.equal(model.output_fc.weight.data[torch.from_numpy(rest).to(DEV)], pretrain[torch.from_numpy(rest).to(DEV)])
    # scoring of the dev codes with the finetuned rows (finetune.py:242): eval mode, no_grad, label subset
    model.eval()
    with torch.no_grad():
        logits, loss = model.forward(x_all[:2], y_all[:2], mask_all[:2], label_indices=syn_t)
    p['output_fc.weight'] = reset
    want = om.gan_model_forward_backward(p, avg, case['x'][:2], case['y'][:2], case['mask'][:2], L, margin, 2.0,
                                         label_indices=syn, need_grad=False)
    close(logits, want['logits'], 1e-5, 2e-5, 'dev scoring logits')
    close(loss, want['loss'], 1e-5, 1e-5, 'dev scoring loss')


def _score_worker(rank, world, port, ref_path, q):
    try:
        sys.path.insert(0, ROOT)
        os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        torch.cuda.set_device(0)
        dist.init_process_group('gloo', rank=rank, world_size=world)
        ref = np.load(ref_path)
        case = _case(4, seed=57)
        model = _gan_model(case, 'cuda')
        model.init_output_fc(torch.from_numpy(ref['fc']).to('cuda'))
        model.enable_label_sharding()
        model.eval()
        x, y, mask = helpers.batch_tensors(case, 'cuda')
        with torch.no_grad():
            logits, loss = model(x, y, mask)
        assert tuple(logits.shape) == (4, L) and not logits.requires_grad
        close(logits, ref['logits'], 1e-5, 2e-5, f'rank {rank}: sharded full-label scoring logits')
        close(loss, ref['loss'], 1e-5, 1e-4, f'rank {rank}: sharded full-label scoring loss')
        torch.cuda.synchronize()
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, 'ok'))
    except Exception:
        q.put((rank, traceback.format_exc()))


def test_sharded_full_label_scoring_at_t2000(tmp_path):
    """Full-label eval scoring (configs[4] "full-label eval scoring at 8 GPUs", here 2 ranks on one GPU): B = 4 notes of
    T = 2000, all 9000 labels, E = H = 200; every rank must hold the oracle's [B, L] logits after the all-gather."""
    case = _case(4, seed=57)
    p = dict(case['params'])
    margin = om.class_margin_from_count(case['target_count'])
    avg, graph, _ = om.label_embedding_fwd(p, case['bufs'])
    rs = np.random.RandomState(58)
    fc = (graph + rs.standard_normal(graph.shape).astype(np.float32) * 0.02).astype(np.float32)
    p['output_fc.weight'] = fc
    X, PX, _ = om.encoder_fwd(p, case['x'], case['mask'], None)
    logits = np.zeros((4, L), np.float32)
    loss = np.float32(0)
    for l0 in range(0, L, 1500):   # chunked: the [B, l, T'] scores of a chunk are 48 MB
        l1 = l0 + 1500
        O, _, _ = om.label_attn_fwd(PX, X, avg[l0:l1])
        lg, ls, _ = om.classifier_loss_fwd(O, p['output_proj.weight'], p['output_proj.bias'], fc[l0:l1], case['y'][:, l0:l1],
                                           margin[l0:l1], 2.0)
        logits[:, l0:l1] = lg
        loss = loss + ls
    ref_path = str(tmp_path / 'ref.npz')
    np.savez(ref_path, logits=logits, loss=np.float32(loss), fc=fc)
    run_ranks(_score_worker, 2, (ref_path,), timeout=300)
