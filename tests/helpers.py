"""Shared helpers for the parity tests (CPU stand-in and GPU)."""
import numpy as np
import torch

from gzsl_text_amd.modules.icd_modules import ConvLabelAttnGAN, ConvLabelAttnModel
from tests.golden import synth


def close(a, b, rtol, atol, what):
    a = np.asarray(a.detach().cpu().numpy() if torch.is_tensor(a) else a, np.float64)
    b = np.asarray(b.detach().cpu().numpy() if torch.is_tensor(b) else b, np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    assert np.isfinite(a).all(), f'{what}: non-finite values'
    err = np.abs(a - b)
    bad = err > atol + rtol * np.abs(b)
    assert not bad.any(), (f'{what}: max err {err.max():.3e} (ref max {np.abs(b).max():.3e}), '
                           f'{int(bad.sum())}/{bad.size} outside rtol={rtol} atol={atol}')


def build_model(case, cls=ConvLabelAttnModel, device='cpu', ops=None, sparse=True, C=2.0):
    return synth.build_model(case, cls=cls, device=device, ops=ops, sparse=sparse, C=C)


batch_tensors = synth.batch_tensors


def check_base_against_golden(model, case, g, train, device='cpu', grad_atol=None):
    d = case['dims']
    x, y, mask = batch_tensors(case, device)
    if train:
        model.train()
        model.set_dropout_seed(synth.DROP_SEED)
    else:
        model.eval()
    model.zero_grad()
    avg, graph = model.get_label_embedding()
    close(avg, g['avg'], 1e-5, 1e-6, 'avg label emb')
    close(graph, g['graph'], 1e-5, 2e-6, 'graph label emb')
    logits, loss = model(x, y, mask)
    loss.mean().backward()
    close(logits, g['logits'], 1e-5, 2e-5, 'logits')
    close(loss, g['loss'], 1e-5, 1e-5, 'loss')
    atol = grad_atol if grad_atol is not None else (2e-6 if d['T'] > 64 else 1e-6)
    for k, p in model.named_parameters():
        close(p.grad, g['grad.' + k], 1e-4, atol, 'grad ' + k)


def check_gan_variant_train_mode(name, g, device='cpu', ops=None):
    """ConvLabelAttnGAN.forward in train mode as finetune.py:258,271 calls it: full label list (seed DROP_SEED), then a
    label_indices subset (the next training forward: DROP_SEED + 1).  logits / loss rtol 1e-5, output_fc gradient rtol
    1e-4 (the tolerances of the eval-mode golden)."""
    case, fc0, sub = synth.gan_train_inputs(name)
    model = build_model(case, cls=ConvLabelAttnGAN, device=device, ops=ops)
    model.load_pretrained_state_dict({k: torch.from_numpy(v.copy()).to(device) for k, v in case['params'].items()})
    model.init_output_fc(torch.from_numpy(fc0).to(device))
    x, y, mask = batch_tensors(case, device)
    model.train()
    model.set_dropout_seed(synth.DROP_SEED)
    model.zero_grad()
    logits, loss = model(x, y, mask)
    loss.backward()
    close(logits, g['full_logits'], 1e-5, 2e-5, 'train-mode full logits')
    close(loss, g['full_loss'], 1e-5, 1e-5, 'train-mode full loss')
    close(model.output_fc.weight.grad, g['full_dfc'], 1e-4, 2e-6, 'train-mode full dfc')
    model.zero_grad()
    sub_t = torch.from_numpy(sub).to(device)
    logits, loss = model(x, y, mask, label_indices=sub_t)
    loss.backward()
    close(logits, g['sub_logits'], 1e-5, 2e-5, 'train-mode subset logits')
    close(loss, g['sub_loss'], 1e-5, 1e-5, 'train-mode subset loss')
    dfc = model.output_fc.weight.grad
    close(dfc[sub_t], g['sub_dfc_rows'], 1e-4, 2e-6, 'train-mode subset dfc rows')
    rest = torch.ones(dfc.shape[0], dtype=torch.bool, device=dfc.device)
    rest[sub_t] = False
    assert not bool(dfc[rest].any()), 'rows outside label_indices must receive no gradient'
    return model
